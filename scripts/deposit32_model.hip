// A model of a 32-bit fixed-point deposit (float canvases): the address patterns of deposit_model.hip with
// ds_add_u32 / ds_add_rtn_u32 on dense rows (all 27 / 64 offsets of a particle are immediates of ONE base address),
// the returned values folded into an overflow guard.  Clocks per wave instruction and CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/deposit32_model.hip -o scripts/deposit32_model && scripts/deposit32_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
struct Cfg {
    int S;        // 3 TSC, 4 PCS
    int pitch;    // row pitch in cells
    int jitter;   // 1: base cell = lattice or lattice - 1, independently per axis and particle
    int swap;     // 1: odd lanes take their second particle first
    int order;    // 0 lattice order (lines of 32 along z), 1 random within the tile
    int mode;     // 0: ds_add_u32; 1: ds_add_rtn_u32, guard = OR of the returned values; 2: ds_add_u64 (8-byte cells)
    int threads;
};
template <int S, int MODE, int PITCH>
__global__ void __launch_bounds__(1024) k(Cfg cf, uint32_t *out, int iters)
{
    extern __shared__ uint32_t lds[];
    constexpr int R0 = 8 + S - 1, R1 = 16 + S - 1, P = PITCH;
    constexpr int cells = R0 * R1 * P * (MODE == 2 ? 2 : 1);
    const int nt = blockDim.x;
    for (int q = threadIdx.x; q < cells; q += nt) lds[q] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t guard = 0;
    for (int it = 0; it < iters; it++) {
        int lb[2][3];
        for (int u = 0; u < 2; u++) {
            const int e = (it * 2 * nt + u * nt + (int)threadIdx.x) & 4095;       // entry of the tile's list
            uint32_t h = hash(e * 7919u + it * 104729u + blockIdx.x * 31u + 17u * u);
            int x, y, z;
            if (cf.order == 0) { z = e & 31; const int L = e >> 5; y = L & 15; x = (L >> 4) & 7; }
            else { z = h & 31; y = (h >> 5) & 15; x = (h >> 9) & 7; h = hash(h); }
            if (cf.jitter) { z -= h & 1; y -= (h >> 1) & 1; x -= (h >> 2) & 1; }
            lb[u][0] = x < 0 ? 0 : x; lb[u][1] = y < 0 ? 0 : y; lb[u][2] = z < 0 ? 0 : z;
        }
        if (cf.swap && (lane & 1)) for (int d = 0; d < 3; d++) { int t = lb[0][d]; lb[0][d] = lb[1][d]; lb[1][d] = t; }
        for (int u = 0; u < 2; u++) {
            const uint32_t v = threadIdx.x + 1 + it;
            const int base = (lb[u][0] * R1 + lb[u][1]) * P + lb[u][2];
#pragma unroll
            for (int a = 0; a < S; a++)
#pragma unroll
                for (int b = 0; b < S; b++)
#pragma unroll
                    for (int c = 0; c < S; c++) {
                        const int idx = base + (a * R1 + b) * P + c;
                        if (MODE == 2) atomicAdd((unsigned long long *)lds + idx, (unsigned long long)v);
                        else if (MODE == 1) guard |= atomicAdd(&lds[idx], v);
                        else atomicAdd(&lds[idx], v);
                    }
        }
    }
    __syncthreads();
    uint32_t s = guard;
    for (int q = threadIdx.x; q < cells; q += nt) s += lds[q];
    if (s == 12345u) out[blockIdx.x] = s;
}
template <int S, int MODE, int PITCH>
static void run(Cfg cf, const char *name)
{
    uint32_t *out; (void)hipMalloc(&out, 1 << 20);
    constexpr int R0 = 8 + S - 1, R1 = 16 + S - 1;
    const size_t lds = (size_t)(R0 * R1 * PITCH) * (MODE == 2 ? 8 : 4);
    int wgs = (int)(160 * 1024 / (lds + 512));
    const int maxw = 2048 / cf.threads;            // 32 waves per CU
    if (wgs > maxw) wgs = maxw;
    const int blocks = 256 * wgs * 4, iters = 100 * 512 / cf.threads;
    auto kern = k<S, MODE, PITCH>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    kern<<<blocks, cf.threads, lds>>>(cf, out, 5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    kern<<<blocks, cf.threads, lds>>>(cf, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * (cf.threads / 64) * iters * 2 * S * S * S;
    printf("%-58s mode %d LDS %5.1f KB x %d wg of %4d: %6.2f clk per wave-instruction\n", name, MODE, lds / 1024.0, wgs, cf.threads,
           (ms * 1e-3) * 2.4e9 * 256 / instr);
    (void)hipFree(out);
}
template <int S, int PITCH> static void trio(int jitter, int swap, int order, int threads, const char *what)
{
    char nm[160];
    snprintf(nm, sizeof nm, "%s pitch %d, %s%s", S == 3 ? "TSC" : "PCS", PITCH, what, swap ? ", swap" : "");
    run<S, 0, PITCH>(Cfg{S, PITCH, jitter, swap, order, 0, threads}, nm);
    run<S, 1, PITCH>(Cfg{S, PITCH, jitter, swap, order, 1, threads}, nm);
    run<S, 2, PITCH>(Cfg{S, PITCH, jitter, swap, order, 2, threads}, nm);
}
int main()
{
    trio<3, 34>(0, 0, 0, 512, "perfect lattice");
    trio<3, 34>(1, 0, 0, 512, "jittered lattice");
    trio<3, 34>(1, 1, 0, 512, "jittered lattice");
    trio<3, 36>(1, 0, 0, 512, "jittered lattice");
    trio<3, 36>(1, 1, 0, 512, "jittered lattice");
    trio<3, 40>(1, 1, 0, 512, "jittered lattice");
    trio<3, 48>(1, 1, 0, 512, "jittered lattice");
    trio<3, 64>(1, 1, 0, 512, "jittered lattice");
    trio<3, 34>(1, 0, 1, 512, "random order in the tile");
    trio<3, 36>(1, 0, 1, 512, "random order in the tile");
    trio<4, 35>(0, 0, 0, 512, "lattice");
    trio<4, 35>(1, 1, 0, 512, "jittered base cells");
    trio<4, 36>(1, 1, 0, 512, "jittered base cells");
    trio<4, 48>(1, 1, 0, 512, "jittered base cells");
    trio<4, 35>(1, 0, 1, 512, "random order in the tile");
    trio<3, 34>(1, 1, 0, 256, "jittered lattice");
    trio<3, 34>(1, 1, 0, 1024, "jittered lattice");
    return 0;
}
