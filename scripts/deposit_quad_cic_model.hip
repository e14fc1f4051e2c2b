// The model of scripts/deposit_quad_model.hip for CIC (S = 2) on a floating-point region (ds_add_f64, what the CIC paint
// uses): one lane per particle (8 atomics per lane) against four lanes per particle (lane q = (a, b), two atomics on the
// z pair), on the split layout of the CIC region (rows of 32 cells + a halo column array), clocks per particle and CU.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/deposit_quad_cic_model.hip -o scripts/deposit_quad_cic_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
constexpr int R0 = 9, R1 = 17, DMAIN = R0 * R1 * 32, CELLS = DMAIN + R0 * R1;
struct Cfg { int mode; int order; int noatom; };       // mode 0 one lane, 1 quad; order 0 lattice 1 jittered 2 random 3 clustered 4 cell-sorted clustered
__device__ __forceinline__ void particle(const Cfg &cf, int e, int it, int *lb, double *d)
{
    uint32_t h = hash(e * 7919u + it * 104729u + blockIdx.x * 31u);
    int x, y, z;
    if (cf.order <= 1) { z = e & 31; const int L = e >> 5; y = L & 15; x = (L >> 4) & 7; if (cf.order == 1) { z -= h & 1; y -= (h >> 1) & 1; x -= (h >> 2) & 1; } }
    else if (cf.order == 2 || (cf.order == 3 && (h & 0x100000))) { z = h & 31; y = (h >> 5) & 15; x = (h >> 9) & 7; }
    else if (cf.order == 3) { z = 13 + (h & 3) % 3; y = 7 + ((h >> 2) & 3) % 3; x = 3 + ((h >> 4) & 3) % 3; }
    else { const int c = e >> 4; z = c & 31; y = (c >> 5) & 7; x = 3; }           // 4: runs of 16 entries per cell (cell-sorted, 16 per cell)
    h = hash(h);
    lb[0] = x < 0 ? 0 : x; lb[1] = y < 0 ? 0 : y; lb[2] = z < 0 ? 0 : z;
    d[0] = (h & 1023) * (1.0 / 1024); d[1] = ((h >> 10) & 1023) * (1.0 / 1024); d[2] = ((h >> 20) & 1023) * (1.0 / 1024);
}
__device__ __forceinline__ int cell(int row, int c) { return c < 32 ? row * 32 + c : DMAIN + row; }
template <int CTRL> __device__ __forceinline__ double qb(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL> __device__ __forceinline__ int qb(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__global__ void __launch_bounds__(512) k(Cfg cf, double *out, int iters)
{
    __shared__ double lds[CELLS];
    for (int i = threadIdx.x; i < CELLS; i += 512) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, q = lane & 3;
    double sink = 0;
    for (int it = 0; it < iters; it++) {
        for (int u = 0; u < 2; u++) {
            int lb[3]; double d[3];
            int e = (it * 1024 + u * 512 + (int)threadIdx.x) & 4095;
            if (cf.mode == 0 && (lane & 1)) e = (it * 1024 + (1 - u) * 512 + (int)threadIdx.x) & 4095;      // odd-lane swap
            particle(cf, e, it, lb, d);
            const double m = 1.0;
            if (cf.mode == 0) {
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++) {
                        const double f = (a ? d[0] : 1 - d[0]) * (b ? d[1] : 1 - d[1]) * m;
                        const int row = (lb[0] + a) * R1 + lb[1] + b;
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            const double v = f * (c ? d[2] : 1 - d[2]);
                            if (cf.noatom) sink += v; else unsafeAtomicAdd(&lds[cell(row, lb[2] + c)], v);
                        }
                    }
            } else {
                const int base = (lb[0] * R1 + lb[1]);
#define SUB(CT) { const double dx = qb<CT>(d[0]), dy = qb<CT>(d[1]), dz = qb<CT>(d[2]); const int rb = qb<CT>(base), z0 = qb<CT>(lb[2]); \
                  const double f = ((q & 2) ? dx : 1 - dx) * ((q & 1) ? dy : 1 - dy) * m; const int row = rb + (q >> 1) * R1 + (q & 1); \
                  const double v0 = f * (1 - dz), v1 = f * dz; \
                  if (cf.noatom) sink += v0 + v1; else { unsafeAtomicAdd(&lds[cell(row, z0)], v0); unsafeAtomicAdd(&lds[cell(row, z0 + 1)], v1); } }
                SUB(0x00) SUB(0x55) SUB(0xaa) SUB(0xff)
#undef SUB
            }
        }
    }
    __syncthreads();
    double s = sink;
    for (int i = threadIdx.x; i < CELLS; i += 512) s += lds[i];
    if (s == 12345.0) out[blockIdx.x] = s;
}
int main()
{
    double *out; (void)hipMalloc(&out, 1 << 20);
    const char *orders[] = {"perfect lattice", "jittered lattice", "random in the tile", "clustered (half in 27 cells)", "cell-sorted, 16 per cell"};
    for (int na = 0; na < 2; na++)
        for (int order = 0; order < 5; order++)
            for (int mode = 0; mode < 2; mode++) {
                Cfg cf{mode, order, na};
                const int blocks = 256 * 4 * 4, iters = 40;
                hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
                k<<<blocks, 512>>>(cf, out, 3); (void)hipDeviceSynchronize();
                (void)hipEventRecord(a); k<<<blocks, 512>>>(cf, out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                printf("%-24s %-30s %s %6.2f clk per particle and CU\n", mode ? "four lanes per particle" : "one lane per particle", orders[order], na ? "NO ATOMICS" : "          ",
                       (ms * 1e-3) * 2.4e9 * 256 / ((double)blocks * iters * 1024));
            }
    return 0;
}
