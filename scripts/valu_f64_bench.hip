// f64 / f32 vector ALU throughput on gfx950 (independent chains, no memory): lanes per clock and CU.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/valu_f64_bench.hip -o valu_bench && ./valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int OP>
__global__ void __launch_bounds__(256) k(T *out, int iters, T a, T b)
{
    T v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (T)(threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) v[i] = v[i] * a;                    // mul
            else if (OP == 1) v[i] = v[i] + b;               // add
            else v[i] = __builtin_fma(v[i], a, b);           // fma
        }
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i];
    if (s == (T)12345) out[0] = s;
}
template <typename T, int OP> void run(const char *name)
{
    T *out; hipMalloc(&out, 1024);
    int blocks = 256 * 16, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<T, OP><<<blocks, 256>>>(out, 10, (T)1.0000001, (T)1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<T, OP><<<blocks, 256>>>(out, iters, (T)1.0000001, (T)1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * 256 * iters * 8;
    printf("%-10s %.3f ms: %.2e lane-ops/s = %.1f lanes/clk/CU\n", name, ms, n / (ms * 1e-3), n / (ms * 1e-3) / 256 / 2.4e9);
}
int main()
{
    run<double, 0>("f64 mul"); run<double, 1>("f64 add"); run<double, 2>("f64 fma");
    run<float, 0>("f32 mul"); run<float, 2>("f32 fma");
    return 0;
}
