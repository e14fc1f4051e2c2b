#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstring>
#include <vector>
struct P28 { double x, y, z; unsigned idx; unsigned pad; };   // 32 bytes
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void fill(int *keys, P28 *vals, unsigned *k2, double *v2, size_t n, int nt) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        keys[i] = (int)(h % nt);
        vals[i] = {1.0 * i, 2.0, 3.0, (unsigned)i, 0};
        k2[i] = (unsigned)(h % n);
        v2[i] = 1.0 * i;
    }
}
template <class K, class V> int run(const char *name, K *kin, K *kout, V *vin, V *vout, size_t n, int b0, int b1) {
    size_t tmp = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, b0, b1));
    void *d; CK(hipMalloc(&d, tmp));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    CK(rocprim::radix_sort_pairs(d, tmp, kin, kout, vin, vout, n, b0, b1));
    CK(hipDeviceSynchronize());
    hipEventRecord(a);
    for (int r = 0; r < 3; r++) CK(rocprim::radix_sort_pairs(d, tmp, kin, kout, vin, vout, n, b0, b1));
    hipEventRecord(b); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s: n=%zu bits [%d,%d) value %zu B: %.3f ms (temp %.1f MB)\n", name, n, b0, b1, sizeof(V), ms / 3, tmp / 1e6);
    hipFree(d);
    return 0;
}
int main() {
    size_t n = 134217728; int nt = 32768;
    int *k, *ko; P28 *v, *vo; unsigned *k2, *k2o; double *v2, *v2o; unsigned *u, *uo;
    CK(hipMalloc(&k, n * 4)); CK(hipMalloc(&ko, n * 4)); CK(hipMalloc(&v, n * sizeof(P28))); CK(hipMalloc(&vo, n * sizeof(P28)));
    CK(hipMalloc(&k2, n * 4)); CK(hipMalloc(&k2o, n * 4)); CK(hipMalloc(&v2, n * 8)); CK(hipMalloc(&v2o, n * 8));
    CK(hipMalloc(&u, n * 4)); CK(hipMalloc(&uo, n * 4));
    fill<<<(unsigned)((n + 255) / 256), 256>>>(k, v, k2, v2, n, nt);
    CK(hipDeviceSynchronize());
    run("tile sort, payload pos+idx", k, ko, v, vo, n, 0, 15);
    run("tile sort, payload idx only", k, ko, (unsigned *)u, (unsigned *)uo, n, 0, 15);
    run("unpermute sort by idx, payload f64", k2, k2o, v2, v2o, n, 0, 27);
    return 0;
}
