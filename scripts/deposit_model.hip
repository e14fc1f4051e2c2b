// A model of tile_deposit's LDS atomics: address patterns of the jittered lattice under TSC / PCS, layouts and lane
// orders, in clocks per wave instruction.   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/deposit_model.hip -o scripts/deposit_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
struct Cfg {
    int S;        // 3 TSC, 4 PCS
    int pitch;    // row pitch in cells; 0 = split layout (rows of 32 + the S-1 halo columns behind)
    int jitter;   // 1: base cell = lattice or lattice - 1, independently per axis and particle
    int swap;     // 1: odd lanes take their second particle first
    int rot;      // 1: every lane walks its z cells in an order rotated by (lane - z) so that the lanes of an instruction hit consecutive banks
    int order;    // 0 lattice order (lines of 32 along z), 1 random within the tile
    int u64;      // 1: ds_add_u64 instead of ds_add_f64
};
template <int S>
__global__ void __launch_bounds__(512) k(Cfg cf, double *out, int iters)
{
    extern __shared__ double lds[];
    const int R0 = 8 + S - 1, R1 = 16 + S - 1;
    const int P = cf.pitch ? cf.pitch : 32;
    const int cells = cf.pitch ? R0 * R1 * P : R0 * R1 * 32 + R0 * R1 * (S - 1);
    const int DMAIN = R0 * R1 * 32;
    for (int q = threadIdx.x; q < cells; q += 512) lds[q] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
        int lb[2][3];
        for (int u = 0; u < 2; u++) {
            const int e = (it * 1024 + u * 512 + (int)threadIdx.x) & 4095;       // entry of the tile's list
            uint32_t h = hash(e * 7919u + it * 104729u + blockIdx.x * 31u + 17u * u);
            int x, y, z;
            if (cf.order == 0) { z = e & 31; const int L = e >> 5; y = L & 15; x = (L >> 4) & 7; }
            else { z = h & 31; y = (h >> 5) & 15; x = (h >> 9) & 7; h = hash(h); }
            if (cf.jitter) { z -= h & 1; y -= (h >> 1) & 1; x -= (h >> 2) & 1; }
            lb[u][0] = x < 0 ? 0 : x; lb[u][1] = y < 0 ? 0 : y; lb[u][2] = z < 0 ? 0 : z;
        }
        if (cf.swap && (lane & 1)) for (int d = 0; d < 3; d++) { int t = lb[0][d]; lb[0][d] = lb[1][d]; lb[1][d] = t; }
        for (int u = 0; u < 2; u++) {
            int r = 0;
            if (cf.rot) {
                // the z of lane 0 of this half wave as reference: lanes then aim at bank (lane + const)
                const int zref = __shfl(lb[u][2], lane & 32);
                r = ((lane & 31) + zref + 1 - lb[u][2]) & 31;
                if (r >= S) r = 0;
            }
            const double v = (double)(threadIdx.x + 1);
#pragma unroll
            for (int a = 0; a < S; a++)
#pragma unroll
                for (int b = 0; b < S; b++) {
                    const int row = (lb[u][0] + a) * R1 + lb[u][1] + b;
#pragma unroll
                    for (int c = 0; c < S; c++) {
                        int cc = c + r; if (cc >= S) cc -= S;
                        const int zc = lb[u][2] + cc;
                        const int idx = cf.pitch ? row * P + zc : (zc < 32 ? row * 32 + zc : DMAIN + row * (S - 1) + (zc - 32));
                        if (cf.u64) atomicAdd((unsigned long long *)&lds[idx], (unsigned long long)threadIdx.x);
                        else unsafeAtomicAdd(&lds[idx], v);
                    }
                }
        }
    }
    __syncthreads();
    double s = 0;
    for (int q = threadIdx.x; q < cells; q += 512) s += lds[q];
    if (s == 12345.0) out[blockIdx.x] = s;
}
static void run(Cfg cf, const char *name)
{
    double *out; (void)hipMalloc(&out, 1 << 20);
    const int S = cf.S, R0 = 8 + S - 1, R1 = 16 + S - 1;
    const size_t lds = (size_t)(cf.pitch ? R0 * R1 * cf.pitch : R0 * R1 * 32 + R0 * R1 * (S - 1)) * 8;
    const int wgs = (int)(160 * 1024 / (lds + 512)) > 4 ? 4 : (int)(160 * 1024 / (lds + 512));
    const int blocks = 256 * wgs * 4, iters = 100;
    auto kern = S == 3 ? k<3> : k<4>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    kern<<<blocks, 512, lds>>>(cf, out, 5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    kern<<<blocks, 512, lds>>>(cf, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 8 * iters * 2 * S * S * S;
    printf("%-74s LDS %5.1f KB x %d: %6.1f clk per wave-instruction\n", name, lds / 1024.0, wgs, (ms * 1e-3) * 2.4e9 * 256 / instr);
    (void)hipFree(out);
}
int main()
{
    //           S pitch jit swap rot order u64
    run(Cfg{3, 0, 0, 0, 0, 0, 0}, "TSC split, perfect lattice");
    run(Cfg{3, 0, 1, 0, 0, 0, 0}, "TSC split, jittered lattice");
    run(Cfg{3, 0, 1, 1, 0, 0, 0}, "TSC split, jittered lattice, swap (the kernel today)");
    run(Cfg{3, 0, 1, 0, 1, 0, 0}, "TSC split, jittered lattice, rotated z order");
    run(Cfg{3, 0, 1, 1, 1, 0, 0}, "TSC split, jittered lattice, swap + rotated z order");
    run(Cfg{3, 0, 1, 0, 0, 1, 0}, "TSC split, random order in the tile");
    run(Cfg{3, 0, 1, 0, 1, 1, 0}, "TSC split, random order in the tile, rotated z order");
    run(Cfg{3, 0, 1, 1, 0, 0, 1}, "TSC split, jittered lattice, swap, ds_add_u64");
    run(Cfg{3, 0, 0, 0, 0, 0, 1}, "TSC split, perfect lattice, ds_add_u64");
    for (int p : {34, 35, 36, 37, 38, 40}) {
        char nm[128];
        snprintf(nm, sizeof nm, "TSC pitch %d, jittered lattice, swap, ds_add_u64", p);
        run(Cfg{3, p, 1, 1, 0, 0, 1}, nm);
        snprintf(nm, sizeof nm, "TSC pitch %d, jittered lattice, no swap, ds_add_u64", p);
        run(Cfg{3, p, 1, 0, 0, 0, 1}, nm);
    }
    run(Cfg{3, 0, 1, 0, 0, 0, 1}, "TSC split, jittered lattice, no swap, ds_add_u64");
    run(Cfg{3, 0, 1, 0, 0, 1, 1}, "TSC split, random order in the tile, ds_add_u64");
    run(Cfg{4, 48, 1, 1, 0, 0, 1}, "PCS pitch 48, jittered base cells, swap, ds_add_u64");
    run(Cfg{4, 48, 1, 0, 0, 0, 1}, "PCS pitch 48, jittered base cells, no swap, ds_add_u64");
    run(Cfg{4, 35, 1, 0, 0, 0, 1}, "PCS pitch 35, jittered base cells, no swap, ds_add_u64");
    run(Cfg{4, 35, 0, 0, 0, 0, 1}, "PCS pitch 35, lattice, ds_add_u64");
    run(Cfg{4, 48, 1, 0, 0, 1, 1}, "PCS pitch 48, random order in the tile, ds_add_u64");
    for (int p : {34, 35, 36, 37, 40, 48}) {
        char nm[128];
        snprintf(nm, sizeof nm, "TSC pitch %d, jittered lattice, swap", p);
        run(Cfg{3, p, 1, 1, 0, 0, 0}, nm);
    }
    run(Cfg{4, 48, 0, 1, 0, 0, 0}, "PCS pitch 48, lattice without jitter, swap (the kernel today)");
    run(Cfg{4, 48, 0, 0, 0, 0, 0}, "PCS pitch 48, lattice without jitter");
    for (int p : {35, 36, 37, 38, 40, 44}) {
        char nm[128];
        snprintf(nm, sizeof nm, "PCS pitch %d, lattice without jitter", p);
        run(Cfg{4, p, 0, 0, 0, 0, 0}, nm);
    }
    run(Cfg{4, 0, 0, 0, 0, 0, 0}, "PCS split, lattice without jitter");
    run(Cfg{4, 48, 1, 1, 0, 0, 0}, "PCS pitch 48, jittered base cells, swap");
    run(Cfg{4, 48, 1, 1, 1, 0, 0}, "PCS pitch 48, jittered base cells, swap + rotated z order");
    run(Cfg{4, 48, 1, 0, 0, 1, 0}, "PCS pitch 48, random order in the tile");
    run(Cfg{4, 48, 0, 0, 0, 0, 1}, "PCS pitch 48, lattice without jitter, ds_add_u64");
    return 0;
}
