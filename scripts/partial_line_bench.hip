// do 128-byte pieces written at unaligned places cost more than aligned ones?  (the bin pass writes its list in
// pieces of ~32 entries = 128 bytes wherever the tile's next free slot happens to be)
//   every half wave writes one piece of 128 bytes; piece p goes to slot perm(p) of a 512 MB buffer, + shift bytes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) pieces(uint32_t *dst, uint32_t npieces_log2, int shift_words, int piece_words)
{
    const uint64_t np = 1ull << npieces_log2;
    const int per = 64 / (128 / 4 * 0 + piece_words);          // pieces per wave
    const uint64_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const uint64_t p = wave * per + lane / piece_words;
    if (p >= np) return;
    const uint64_t slot = (p * 0x9E3779B1ull) & (np - 1);       // odd multiplier: a permutation of the slots
    dst[slot * piece_words + shift_words + lane % piece_words] = (uint32_t)p;
}
int main()
{
    const uint32_t lg = 22;                                     // 4 M pieces of 128 bytes = 512 MB
    uint32_t *d; hipMalloc(&d, ((size_t)128 << lg) + 4096);
    hipMemset(d, 0, ((size_t)128 << lg) + 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int piece_words : {32, 64}) for (int shift : {0, 4, 12, 16}) {
        const uint64_t np = (piece_words == 32) ? (1ull << lg) : (1ull << (lg - 1));
        const uint32_t l2 = piece_words == 32 ? lg : lg - 1;
        const int per = 64 / piece_words;
        const unsigned blocks = (unsigned)((np / per * 64 + 255) / 256);
        pieces<<<blocks, 256>>>(d, l2, shift, piece_words); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 10; r++) pieces<<<blocks, 256>>>(d, l2, shift, piece_words);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
        printf("pieces of %3d bytes, shifted by %2d bytes: %.3f ms  %.2f TB/s written\n", piece_words * 4, shift * 4, ms, (double)np * piece_words * 4 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
