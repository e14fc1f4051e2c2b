// LDS atomic throughput on gfx950 by operand type: lanes per clock and CU for ds_add_{u32,u64,f32,f64}
// with a distinct address per lane (no same-address serialisation), 27 atomics per "particle" like TSC.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/ldsatomic_bench.hip -o /tmp/ldsatomic && /tmp/ldsatomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T> __device__ void add(T *p, T v) { atomicAdd(p, v); }
template <> __device__ void add<double>(double *p, double v) { unsafeAtomicAdd(p, v); }
template <> __device__ void add<float>(float *p, float v) { unsafeAtomicAdd(p, v); }

template <typename T, int STRIDE>
__global__ void __launch_bounds__(256) k(T *out, int iters)
{
    __shared__ T lds[4096 + 64];
    for (int q = threadIdx.x; q < 4096 + 64; q += 256) lds[q] = 0;
    __syncthreads();
    T v = (T)(threadIdx.x + 1);
    int base = (threadIdx.x * STRIDE) & 4095;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < 27; c++) add<T>(&lds[(base + c * 37 + it) & 4095], v);
    }
    __syncthreads();
    T s = 0;
    for (int q = threadIdx.x; q < 4096; q += 256) s += lds[q];
    if (s == (T)12345) out[blockIdx.x] = s;
}

template <typename T, int STRIDE> void run(const char *name)
{
    T *out; hipMalloc(&out, 1 << 20);
    int blocks = 256 * 8, iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<T, STRIDE><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<T, STRIDE><<<blocks, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double)blocks * 256 * iters * 27;
    printf("%-22s stride %d: %.3f ms, %.2e atomics/s = %.2f lanes/clk/CU (256 CUs, 2.4 GHz)\n", name, STRIDE, ms, n / (ms * 1e-3),
           n / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(out);
}

int main()
{
    run<uint32_t, 1>("ds_add_u32");
    run<unsigned long long, 1>("ds_add_u64");
    run<float, 1>("ds_add_f32");
    run<double, 1>("ds_add_f64");
    run<uint32_t, 3>("ds_add_u32");
    run<unsigned long long, 3>("ds_add_u64");
    run<double, 3>("ds_add_f64");
    return 0;
}
