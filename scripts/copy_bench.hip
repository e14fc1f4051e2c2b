// what does a streaming copy reach on this box, and how does it depend on the launch shape?
//   grid-stride (all workgroups sweep the array together) vs chunked (each workgroup owns a contiguous piece),
//   workgroups per launch, threads per workgroup, nontemporal hints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float __attribute__((ext_vector_type(4))) f4;
template <int NT, int UN, int TH, int CHUNKED>
__global__ void __launch_bounds__(TH) copyk(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    size_t i0, i1, stride;
    if (CHUNKED) { const size_t per = (n + gridDim.x - 1) / gridDim.x; i0 = blockIdx.x * per + threadIdx.x; i1 = min(n, (blockIdx.x + 1) * per); stride = TH; }
    else { i0 = blockIdx.x * (size_t)TH + threadIdx.x; i1 = n; stride = (size_t)gridDim.x * TH; }
    for (size_t i = i0; i < i1; i += stride * UN) {
        f4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) { const size_t j = i + u * stride; if (j < i1) v[u] = (NT & 1) ? __builtin_nontemporal_load(&src[j]) : src[j]; }
#pragma unroll
        for (int u = 0; u < UN; u++) { const size_t j = i + u * stride; if (j < i1) { if (NT & 2) __builtin_nontemporal_store(v[u], &dst[j]); else dst[j] = v[u]; } }
    }
}
template <int NT, int UN, int TH, int CHUNKED> void run(const f4 *s, f4 *d, size_t n, int blocks)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    copyk<NT, UN, TH, CHUNKED><<<blocks, TH>>>(s, d, n);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 20; r++) copyk<NT, UN, TH, CHUNKED><<<blocks, TH>>>(s, d, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%s nt %d unroll %d threads %d blocks %6d: %.3f ms  %.2f TB/s\n", CHUNKED ? "chunked    " : "grid-stride", NT, UN, TH, blocks, ms, 2.0 * n * 16 / (ms * 1e-3) / 1e12);
}
int main()
{
    const size_t cap = (size_t)1 << 30;
    f4 *s, *d; hipMalloc(&s, cap); hipMalloc(&d, cap); hipMemset(s, 1, cap); hipMemset(d, 0, cap);
    // working set = source + destination
    for (size_t mb : {8, 16, 32, 64, 128, 192, 256, 512, 2048}) {
        const size_t n = (mb << 20) / 32;
        printf("working set %zu MB\n", mb);
        for (int blocks : {1024, 2048, 4096}) { run<0, 1, 256, 0>(s, d, n, blocks); run<0, 2, 256, 0>(s, d, n, blocks); run<0, 1, 1024, 0>(s, d, n, blocks); }
    }
    return 0;
}
