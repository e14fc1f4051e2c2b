// Which lanes of a wave meet in the LDS when a 64-bit atomic is issued?  (Round 4: the S >= 3 paint kernels are
// bound by ds_add_u64 at ~11.5 clocks per wave instruction on the benchmark's jittered lattice against 6.3 for a
// conflict-free one.  If the hardware serves the instruction in fixed groups of lanes, a lane <-> list entry
// permutation that keeps neighbouring entries — the ones the jitter puts on one cell — out of each other's group
// costs nothing and removes the conflicts.)
//   hipcc --offload-arch=gfx950 -O3 scripts/ldsatomic_groups.hip -o scripts/ldsatomic_groups
// Patterns (cell = 8-byte word of a 2 x 32-cell pair of rows, rows 256 bytes apart unless PITCH says otherwise):
//   linear        lane l -> row l / 32, cell l % 32                  (conflict free: the floor)
//   pair d        lanes l and l ^ d share an ADDRESS                 (d = 1, 2, 4, 8, 16, 32)
//   bank d        lanes l and l ^ d share a BANK PAIR, different rows (d = 16, 32: only for d >= 32 natural)
//   jitter        lane l -> cell l % 32 + delta_l, delta in {-1, 0} pseudo-random (the benchmark's lattice)
//   jitter-perm   the same cells dealt to the lanes as [even entries of row A | odd of A | even of B | odd of B]
//   jitter-perm32 ... as [even of A, odd of B | odd of A, even of B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int CELLS = 4096;
enum { LINEAR, PAIR, BANK, JITTER, JPERM16, JPERM32, JPERM8 };
__device__ __forceinline__ int jitter_cell(int entry, uint32_t salt)
{
    // entry 0..63: row entry / 32, cell entry % 32 - (0 or 1), never below 0 in the row (cell 0 stays)
    uint32_t h = (uint32_t)(entry + 1) * 2654435761u ^ salt;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    int z = entry % 32 - (int)(h & 1);
    if (z < 0) z = 0;
    return (entry / 32) * 32 + z;
}
template <int OP>      // 0: ds_add_u64, 1: ds_add_f64, 2: ds_add_u32
__global__ void __launch_bounds__(512) k(double *out, int iters, int pattern, int d, int pitch)
{
    __shared__ double lds[CELLS];
    for (int q = threadIdx.x; q < CELLS; q += 512) lds[q] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; it++) {
        int cell;
        const uint32_t salt = (uint32_t)(it * 977 + wave * 131 + blockIdx.x);
        if (pattern == LINEAR) cell = (lane / 32) * pitch + lane % 32;
        else if (pattern == PAIR) { int l = lane & ~d; cell = (l / 32) * pitch + l % 32; }
        else if (pattern == BANK) { int l = lane & ~d; cell = (l / 32) * pitch + l % 32 + ((lane & d) ? 2 * pitch : 0); }
        else {
            int entry = lane;
            if (pattern == JPERM16) entry = (lane & 32) | ((lane & 15) << 1) | ((lane >> 4) & 1);
            if (pattern == JPERM32) {
                // lanes 0-15: even of A, 16-31: odd of B, 32-47: odd of A, 48-63: even of B
                const int q = lane >> 4, k2 = (lane & 15) << 1;
                entry = q == 0 ? k2 : (q == 1 ? 32 + k2 + 1 : (q == 2 ? k2 + 1 : 32 + k2));
            }
            if (pattern == JPERM8) entry = ((lane & 7) << 3) | (lane >> 3);      // 8 x 8 transpose
            const int c = jitter_cell(entry, salt);
            cell = (c / 32) * pitch + c % 32;
        }
        cell += ((wave * 7 + it) & 7) * 4 * pitch;            // (different waves work on different rows)
#pragma unroll
        for (int c = 0; c < 9; c++) {
            // 9 x 3 stencil points: row offsets c, cell offsets 0..2 (immediate offsets)
#pragma unroll
            for (int e = 0; e < 3; e++) {
                double *p = &lds[(cell + c * 8 * pitch + e) % CELLS];
                if (OP == 0) atomicAdd((unsigned long long *)p, (unsigned long long)threadIdx.x);
                else if (OP == 1) unsafeAtomicAdd(p, (double)threadIdx.x);
                else atomicAdd((unsigned int *)p, (unsigned int)threadIdx.x);
            }
        }
    }
    __syncthreads();
    double s = 0;
    for (int q = threadIdx.x; q < CELLS; q += 512) s += lds[q];
    if (s == 12345.0) out[blockIdx.x] = s;
}
template <int OP> void run(const char *name, int pattern, int d, int pitch)
{
    double *out; hipMalloc(&out, 1 << 20);
    int blocks = 256 * 3 * 4, iters = 200;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 512>>>(out, 10, pattern, d, pitch);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 512>>>(out, iters, pattern, d, pitch);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr = (double)blocks * 8 * iters * 27;
    printf("%-12s %-22s d %2d pitch %2d: %.3f ms: %5.1f clk per wave-instruction per CU\n", OP == 0 ? "ds_add_u64" : (OP == 1 ? "ds_add_f64" : "ds_add_u32"),
           name, d, pitch, ms, (ms * 1e-3) * 2.4e9 * 256 / instr);
    hipFree(out);
}
template <int OP> void all()
{
    for (int pitch : {32, 34, 48}) {
        run<OP>("linear", LINEAR, 0, pitch);
        for (int d : {1, 2, 4, 8, 16, 32}) run<OP>("same address, lanes l^d", PAIR, d, pitch);
        for (int d : {1, 2, 4, 8, 16, 32}) run<OP>("same bank, lanes l^d", BANK, d, pitch);
        run<OP>("jitter", JITTER, 0, pitch);
        run<OP>("jitter perm16", JPERM16, 0, pitch);
        run<OP>("jitter perm32", JPERM32, 0, pitch);
        run<OP>("jitter perm8x8", JPERM8, 0, pitch);
    }
}
int main()
{
    all<0>();
    all<1>();
    all<2>();
    return 0;
}
