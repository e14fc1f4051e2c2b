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
    return 0;
}
