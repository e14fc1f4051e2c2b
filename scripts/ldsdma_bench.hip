// cost of LDS-DMA gathers (24-byte rows, L2 resident) and of the flush pattern of the walk kernels in
// clocks per wave-instruction, all 8 waves of a 512-thread workgroup per CU issuing
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int BYTES> __device__ __forceinline__ void glds(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    if constexpr (BYTES == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// MODE 0: x4 + dword + dword DMA per row (the f8 record); 1: dwordx4 only; 2: dword only x3;
// 3: register loads (dwordx4 + dwordx2) + ds_write; 4: flush pattern 9 ds_add_f64; 5: 9 ds_add_f64 same row
template <int MODE>
__global__ void __launch_bounds__(512) k(const char *src, int nrows, int iters, unsigned long long *out, double *sink)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t base = (uint32_t)(uintptr_t)smem;
    double *ring = (double *)(smem + 65536);
    for (int i = tid; i < 4 * 18 * 34; i += 512) ring[i] = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double accv = 1.0;
    for (int it = 0; it < iters; it++) {
        const int row = (it * 512 + tid) % nrows;
        const char *r = src + (size_t)row * 24;
        if (MODE == 0) {
            glds<16>(r, base + wave * 64 * 16); glds<4>(r + 16, base + 32768 + wave * 256); glds<4>(r + 20, base + 40960 + wave * 256);
            if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 1) {
            glds<16>(r, base + wave * 64 * 16);
            if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 2) {
            glds<4>(r, base + wave * 256); glds<4>(r + 16, base + 32768 + wave * 256); glds<4>(r + 20, base + 40960 + wave * 256);
            if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 3) {
            const double2 xy = *(const double2 *)r; const double z = *(const double *)(r + 16);
            *(double2 *)(smem + tid * 16) = xy; *(double *)(smem + 32768 + tid * 8) = z;
        } else if (MODE == 4) {
            const int tb = tid / 32, tc = tid % 32;
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int kk = 0; kk < 3; kk++) unsafeAtomicAdd(&ring[(it & 3) * 612 + (tb + j) * 34 + tc + kk], accv);
        } else {
            const int tb = tid / 32, tc = tid % 32;
#pragma unroll
            for (int j = 0; j < 9; j++) unsafeAtomicAdd(&ring[(j & 3) * 612 + tb * 34 + tc], accv);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (sink && tid == 12345) sink[0] = ring[tid] + smem[tid];
}
template <int MODE> void run(const char *name, int per_iter)
{
    const int nrows = 1 << 16, iters = 2000, nb = 256;
    char *src; unsigned long long *out;
    hipMalloc(&src, (size_t)nrows * 24 + 64); hipMemset(src, 1, (size_t)nrows * 24 + 64);
    hipMalloc(&out, nb * 8);
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    k<MODE><<<nb, 512, 98304>>>(src, nrows, iters, out, nullptr);
    k<MODE><<<nb, 512, 98304>>>(src, nrows, iters, out, nullptr);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, out, nb * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < nb; i++) s += h[i];
    s /= nb;
    printf("%-40s %8.1f clocks per iteration of the workgroup = %.1f per wave-instruction (x8 waves x%d)\n", name, s / iters, s / iters / (8.0 * per_iter), per_iter);
    hipFree(src); hipFree(out);
}
int main()
{
    run<0>("DMA x4 + dword + dword (24 B row)", 3);
    run<1>("DMA x4 only", 1);
    run<2>("DMA dword x3", 3);
    run<3>("register loads + ds_write (24 B)", 2);
    run<4>("flush: 9 ds_add_f64, 3x3 stencil", 9);
    run<5>("9 ds_add_f64, own cell", 9);
    return 0;
}
