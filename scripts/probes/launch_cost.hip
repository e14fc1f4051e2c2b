// host cost of hipLaunchKernel / hipMemsetAsync on an idle and on a busy stream (what the one-rank cycle's issue time is made of)
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
struct Big { double a[40]; long b[12]; };
__global__ void tiny(uint32_t *p, Big g) { if (threadIdx.x == 0 && blockIdx.x == 0 && g.a[0] == 12345.0) p[0] = 1; }
__global__ void zero(uint32_t *p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0; }
__global__ void busy(uint32_t *p, int iters) { uint32_t x = threadIdx.x; for (int i = 0; i < iters; i++) x = x * 1664525u + 1013904223u; if (x == 42) p[0] = x; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st; (void)hipStreamCreate(&st);
    uint32_t *d; (void)hipMalloc(&d, 1 << 20);
    Big g{}; 
    for (int mode = 0; mode < 6; mode++) {
        // 0: 4 launches; 1: memset + 3 launches; 2: zero kernel + 3 launches; 3/4/5: the same behind a 200 us kernel
        const int K = 300;
        double tm = 0, tl = 0, worst = 0;
        for (int k = 0; k < K + 20; k++) {
            if (k == 20) { tm = tl = worst = 0; }
            if (mode >= 3) busy<<<256, 256, 0, st>>>(d, 60000);
            double t0 = now();
            if (mode % 3 == 1) (void)hipMemsetAsync(d, 0, 20000, st);
            else if (mode % 3 == 2) zero<<<16, 256, 0, st>>>(d, 5000);
            else tiny<<<1, 64, 0, st>>>(d, g);
            double t1 = now();
            for (int j = 0; j < 3; j++) tiny<<<64, 256, 0, st>>>(d, g);
            double t2 = now();
            tm += t1 - t0; tl += (t2 - t1) / 3; if (t2 - t1 > worst) worst = t2 - t1;
            (void)hipStreamSynchronize(st);
        }
        printf("mode %d (%s%s): first op %.2f us, each following launch %.2f us (worst trio %.1f us)\n", mode,
               mode % 3 == 1 ? "hipMemsetAsync" : mode % 3 == 2 ? "zero kernel" : "launch", mode >= 3 ? ", behind a running kernel" : "", tm / K, tl / K, worst);
    }
    return 0;
}
