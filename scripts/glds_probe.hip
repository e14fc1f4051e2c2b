// semantics of global_load_lds_{dword,dwordx3,dwordx4} on gfx950: where do the lanes' bytes land,
// are inactive lanes skipped, does an LDS base above 64 KB work?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
template <int BYTES> __device__ __forceinline__ void glds(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    if constexpr (BYTES == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else if constexpr (BYTES == 12)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int BYTES>
__global__ void probe(const uint32_t *src, uint32_t *out, int stride_bytes, int nactive, uint32_t ldsoff, int total_dwords)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *w = (uint32_t *)smem;
    for (int i = threadIdx.x; i < total_dwords; i += blockDim.x) w[i] = 0xdeadbeefu;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t base = (uint32_t)(uintptr_t)smem + ldsoff + wave * 64 * BYTES;
    if (lane < nactive) glds<BYTES>((const char *)src + (size_t)(threadIdx.x) * stride_bytes, base);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < total_dwords; i += blockDim.x) out[i] = w[i];
}
template <int BYTES> void run(int stride, int nactive, uint32_t ldsoff, int ldsbytes)
{
    const int nt = 128;
    std::vector<uint32_t> h(1 << 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)i;
    uint32_t *src, *out;
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, ldsbytes);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)probe<BYTES>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsbytes);
    probe<BYTES><<<1, nt, ldsbytes>>>(src, out, stride, nactive, ldsoff, ldsbytes / 4);
    hipError_t e = hipDeviceSynchronize();
    std::vector<uint32_t> o(ldsbytes / 4);
    hipMemcpy(o.data(), out, ldsbytes, hipMemcpyDeviceToHost);
    printf("BYTES %d stride %d nactive %d ldsoff %u: %s\n", BYTES, stride, nactive, ldsoff, hipGetErrorString(e));
    // expected: lane t of wave w writes BYTES/4 dwords src[(t*stride)/4 + k] at ldsoff + (w*64 + lane)*BYTES
    int bad = 0, written = 0;
    for (int t = 0; t < nt; t++) {
        const int lane = t & 63;
        for (int k = 0; k < BYTES / 4; k++) {
            const uint32_t got = o[ldsoff / 4 + t * (BYTES / 4) + k];
            const uint32_t want = lane < nactive ? (uint32_t)(t * stride / 4 + k) : 0xdeadbeefu;
            if (got != want) { if (bad < 6) printf("  t %d k %d got %08x want %08x\n", t, k, got, want); bad++; }
        }
    }
    for (size_t i = 0; i < o.size(); i++) written += o[i] != 0xdeadbeefu;
    printf("  mismatches %d, dwords written anywhere %d (expected %d)\n", bad, written, 2 * nactive * BYTES / 4);
    hipFree(src); hipFree(out);
}
int main()
{
    run<4>(24, 64, 0, 4096); run<4>(24, 5, 0, 4096);
    run<16>(24, 64, 0, 8192); run<16>(24, 7, 1024, 8192);
    run<12>(12, 64, 0, 8192); run<12>(12, 9, 0, 8192);
    run<4>(4, 64, 70000, 81920); run<16>(16, 64, 98304, 131072);
    return 0;
}
