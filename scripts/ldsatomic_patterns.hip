// ds_add_f64 / ds_write_b64 / ds_read_b64 rate on gfx950 under the ADDRESS PATTERNS of the tile kernels:
// a 10 x 18 x 34 region (TSC), 27 stencil points per "particle", base cells that are
//   0 lane-linear (lane <-> c: conflict free)            1 uniformly random in the tile
//   2 lane-linear with every other lane sharing its left neighbour's cell (same-address pairs)
//   3 random within a 5^3 blob
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/ldsatomic_patterns.hip -o scripts/ldsatomic_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int R0 = 10, R1 = 18, R2 = 34, CELLS = R0 * R1 * R2;
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int PATTERN> __device__ __forceinline__ int base_cell(int tid, int it)
{
    const int lane = tid & 63, wave = tid >> 6;
    if (PATTERN == 0) { int c = lane & 31, b = (wave * 2 + (lane >> 5) + it) & 15, a = (it >> 4) & 7; return (a * R1 + b) * R2 + c; }
    if (PATTERN == 2) { int l2 = lane & ~1; int c = l2 & 31, b = (wave * 2 + (l2 >> 5) + it) & 15, a = (it >> 4) & 7; return (a * R1 + b) * R2 + c; }
    uint32_t h = hash(tid * 7919u + it * 104729u);
    if (PATTERN == 1) { int c = h & 31, b = (h >> 5) & 15, a = (h >> 9) & 7; return (a * R1 + b) * R2 + c; }
    int c = 10 + (h % 5), b = 6 + ((h >> 8) % 5), a = 2 + ((h >> 16) % 5); return (a * R1 + b) * R2 + c;
}
// OP 0: ds_add_f64 / ds_add_f32, 1: ds_write, 2: ds_read;  T: the element type of the region
template <int PATTERN, int OP, typename T>
__global__ void __launch_bounds__(512) k(double *out, int iters)
{
    __shared__ T lds[CELLS];
    for (int q = threadIdx.x; q < CELLS; q += 512) lds[q] = 0;
    __syncthreads();
    T v = (T)(threadIdx.x + 1), acc = 0;
    for (int it = 0; it < iters; it++) {
        const int base = base_cell<PATTERN>(threadIdx.x, it);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    T *p = &lds[base + (a * R1 + b) * R2 + c];
                    if (OP == 0) unsafeAtomicAdd(p, v);
                    else if (OP == 1) *(volatile T *)p = v;
                    else acc += *(volatile T *)p;
                }
    }
    __syncthreads();
    double s = (double)acc;
    for (int q = threadIdx.x; q < CELLS; q += 512) s += lds[q];
    if (s == 12345.0) out[blockIdx.x] = s;
}
template <int PATTERN, int OP, typename T = double> void run(const char *name)
{
    double *out; hipMalloc(&out, 1 << 20);
    int blocks = 256 * 3 * 4, iters = 200;       // 3 workgroups of 512 per CU resident (49 KB each), 4 rounds
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<PATTERN, OP, T><<<blocks, 512>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<PATTERN, OP, T><<<blocks, 512>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double)blocks * 512 * iters * 27;
    printf("%-44s %.3f ms: %.2f lanes/clk/CU = %.1f clk per wave-instruction (256 CUs at 2.4 GHz)\n", name, ms,
           n / (ms * 1e-3) / 256 / 2.4e9, 64.0 / (n / (ms * 1e-3) / 256 / 2.4e9));
    hipFree(out);
}

// ---- the jittered lattice of the benchmark under TSC: lane <-> z cell, but every axis' base cell is the
// lattice cell or the one before it (floor(x + 0.5) with x within +-0.4 of a cell centre), independently per
// particle.  RP: row pitch in cells (34 = the kernel's T2 + 2; 48 = 384 bytes, a multiple of the 128-byte bank
// row: then the x / y jitter moves no lane to another bank), PP: plane pitch in rows.
template <int RP, int PP, int JX, int JZ>
__global__ void __launch_bounds__(512) kj(double *out, int iters)
{
    extern __shared__ double ldsj[];
    constexpr int CELLSJ = 10 * PP * RP;
    for (int q = threadIdx.x; q < CELLSJ; q += 512) ldsj[q] = 0;
    __syncthreads();
    double v = (double)(threadIdx.x + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; it++) {
        const uint32_t h = hash(threadIdx.x * 7919u + it * 104729u + blockIdx.x * 31u);
        int c = 1 + (lane & 31) - (JZ ? (h & 1) : 0), b = 1 + ((wave * 2 + (lane >> 5) + it) & 7) * 2 - (JX ? ((h >> 1) & 1) : 0),
            a = 1 + ((it >> 3) & 3) * 2 - (JX ? ((h >> 2) & 1) : 0);
        if (c > 31) c = 31;
        const int base = (a * PP + b) * RP + c;
#pragma unroll
        for (int a2 = 0; a2 < 3; a2++)
#pragma unroll
            for (int b2 = 0; b2 < 3; b2++)
#pragma unroll
                for (int c2 = 0; c2 < 3; c2++) unsafeAtomicAdd(&ldsj[base + (a2 * PP + b2) * RP + c2], v);
    }
    __syncthreads();
    double s = 0;
    for (int q = threadIdx.x; q < CELLSJ; q += 512) s += ldsj[q];
    if (s == 12345.0) out[blockIdx.x] = s;
}
template <int RP, int PP, int JX, int JZ> void runj(const char *name, int wgs_per_cu)
{
    double *out; hipMalloc(&out, 1 << 20);
    const int blocks = 256 * wgs_per_cu * 4, iters = 200;
    const size_t lds = (size_t)10 * PP * RP * 8;
    hipFuncSetAttribute((const void *)kj<RP, PP, JX, JZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kj<RP, PP, JX, JZ><<<blocks, 512, lds>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kj<RP, PP, JX, JZ><<<blocks, 512, lds>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double)blocks * 512 * iters * 27;
    printf("%-64s LDS %5.1f KB: %.3f ms: %.1f clk per wave-instruction\n", name, lds / 1024.0, ms, 64.0 / (n / (ms * 1e-3) / 256 / 2.4e9));
    hipFree(out);
}
int main()
{
    run<0, 0>("ds_add_f64  lane-linear");
    run<1, 0>("ds_add_f64  random in the tile");
    run<2, 0>("ds_add_f64  lane-linear, same-cell pairs");
    run<3, 0>("ds_add_f64  random in a 5^3 blob");
    run<0, 1>("ds_write_b64 lane-linear");
    run<1, 1>("ds_write_b64 random in the tile");
    run<0, 2>("ds_read_b64 lane-linear");
    run<1, 2>("ds_read_b64 random in the tile");
    run<3, 2>("ds_read_b64 random in a 5^3 blob");
    run<0, 0, float>("ds_add_f32  lane-linear");
    run<1, 0, float>("ds_add_f32  random in the tile");
    run<2, 0, float>("ds_add_f32  lane-linear, same-cell pairs");
    run<3, 0, float>("ds_add_f32  random in a 5^3 blob");
    run<0, 1, float>("ds_write_b32 lane-linear");
    run<1, 2, float>("ds_read_b32 random in the tile");
    runj<34, 18, 0, 0>("TSC lattice, no jitter, pitch 34 x 18", 3);
    runj<34, 18, 1, 0>("TSC lattice, x / y jitter, pitch 34 x 18", 3);
    runj<34, 18, 0, 1>("TSC lattice, z jitter, pitch 34 x 18", 3);
    runj<34, 18, 1, 1>("TSC lattice, x / y / z jitter, pitch 34 x 18 (the kernel today)", 3);
    runj<48, 18, 1, 0>("TSC lattice, x / y jitter, pitch 48 x 18 (rows = 3 bank rows)", 2);
    runj<48, 18, 1, 1>("TSC lattice, x / y / z jitter, pitch 48 x 18", 2);
    runj<34, 18, 1, 1>("TSC lattice, x / y / z jitter, pitch 34 x 18, 2 workgroups per CU", 2);
    runj<36, 18, 1, 1>("TSC lattice, x / y / z jitter, pitch 36 x 18", 3);
    runj<40, 18, 1, 1>("TSC lattice, x / y / z jitter, pitch 40 x 18", 2);
    runj<34, 19, 1, 1>("TSC lattice, x / y / z jitter, pitch 34 x 19", 3);
    return 0;
}
