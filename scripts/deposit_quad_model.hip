// A model of the S = 4 (PCS) deposit into a 64-bit fixed-point LDS region, WITH its arithmetic (weights, products, the
// fixed-point conversion), in two formulations:
//   mode 0  one lane per particle (paint_tile_kernel today): 64 ds_add_u64 per lane, the 64 lanes of an instruction are 64
//           different particles — neighbours on the lattice, which meet on cells and banks;
//   mode 1-3  four lanes per particle ("quad"): lane q of a quad owns stencil index q along one axis (1: x, 2: y, 3: z) and
//           walks the 16 points of the other two; the 64 lanes of an instruction are 16 particles x 4 indices.  A quad's
//           particle reaches its lanes by DPP quad_perm moves (no LDS, no readlane); each lane evaluates the one weight of
//           its own index from per-lane polynomial coefficients and the 2 x 4 weights of the other axes as today.
// Question: does a formulation with fewer DIFFERENT particles per LDS-atomic instruction pay less
// for the conflicts of the jittered lattice / of a clustered set than it adds in vector work?
// Reported: clocks per particle and CU (2.4 GHz, 256 CUs), i.e. what the deposit loop of a PCS paint costs.
//   hipcc --offload-arch=gfx950 -O3 scripts/deposit_quad_model.hip -o scripts/deposit_quad_model && scripts/deposit_quad_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
constexpr int S = 4, R0 = 8 + S - 1, R1 = 16 + S - 1;
constexpr double MAGIC = 6755399441055744.0;
constexpr long long MAGIC_BITS = 0x4338000000000000ll;
struct Cfg {
    int mode;     // 0 lane per particle; 1 / 2 / 3 quad along x / y / z
    int order;    // 0 lattice order, 1 jittered lattice, 2 random within the tile, 3 clustered (half of the particles in a 3^3 block of cells)
    int swap;     // mode 0: odd lanes take their second particle first (the kernel today)
    int noatom;   // 1: the arithmetic alone (the sum goes to a register)
};
__device__ __forceinline__ void pcs(double d, double *V)
{
    const double e = d - 1.0, d2 = d * d, e2 = e * e;
    V[1] = __builtin_fma(d2, __builtin_fma(0.5, d, -1.0), 2.0 / 3.0);
    V[2] = __builtin_fma(e2, __builtin_fma(-0.5, e, -1.0), 2.0 / 3.0);
    V[0] = e2 * (e * (-1.0 / 6.0));
    V[3] = d2 * (d * (1.0 / 6.0));
}
// the synthetic particle `e` of tile-list trip `it`: base cell in the tile and offsets in the cell
__device__ __forceinline__ void particle(const Cfg &cf, int e, int it, int *lb, double *d)
{
    uint32_t h = hash(e * 7919u + it * 104729u + blockIdx.x * 31u);
    int x, y, z;
    if (cf.order <= 1) { z = e & 31; const int L = e >> 5; y = L & 15; x = (L >> 4) & 7; if (cf.order == 1) { z -= h & 1; y -= (h >> 1) & 1; x -= (h >> 2) & 1; } }
    else if (cf.order == 2 || (h & 0x100000)) { z = h & 31; y = (h >> 5) & 15; x = (h >> 9) & 7; }
    else { z = 13 + (h & 3) % 3; y = 7 + ((h >> 2) & 3) % 3; x = 3 + ((h >> 4) & 3) % 3; }
    h = hash(h);
    lb[0] = x < 0 ? 0 : x; lb[1] = y < 0 ? 0 : y; lb[2] = z < 0 ? 0 : z;
    d[0] = (h & 1023) * (1.0 / 1024); d[1] = ((h >> 10) & 1023) * (1.0 / 1024); d[2] = ((h >> 20) & 1023) * (1.0 / 1024);
}
template <int CTRL> __device__ __forceinline__ double quad_bcast(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL> __device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

template <int PITCH, int AX>
__device__ __forceinline__ void quad_subtrip(const Cfg &cf, double *lds, const double *dd, int base, int q, const double *co, double ms, double &sink)
{
    // dd / base: this sub-trip's particle (already broadcast within the quad)
    double Wa[4], Wb[4];
    constexpr int A1 = AX == 0 ? 1 : 0, A2 = AX == 2 ? 1 : 2;          // the two walked axes (A1 < A2)
    pcs(dd[A1], Wa); pcs(dd[A2], Wb);
    const double dq = dd[AX];
    const double wown = __builtin_fma(__builtin_fma(__builtin_fma(co[3], dq, co[2]), dq, co[1]), dq, co[0]) * ms;
    constexpr int STRIDE[3] = {R1 * PITCH, PITCH, 1};
    const int mine = base + q * STRIDE[AX];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double f = wown * Wa[i];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double r = __builtin_fma(f, Wb[j], MAGIC);
            const int cell = mine + i * STRIDE[A1] + j * STRIDE[A2];
            if (cf.noatom) sink += r;
            else atomicAdd((unsigned long long *)&lds[cell], (unsigned long long)(__double_as_longlong(r) - MAGIC_BITS));
        }
    }
}

template <int PITCH, int AX>
__global__ void __launch_bounds__(512) k(Cfg cf, double *out, int iters)
{
    extern __shared__ double lds[];
    constexpr int cells = R0 * R1 * PITCH;
    for (int i = threadIdx.x; i < cells; i += 512) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double sink = 0;
    const double ms = 1.0 * 1125899906842624.0 / 16384;     // mass x 2^f
    if (cf.mode == 0) {
        for (int it = 0; it < iters; it++) {
            int lb[2][3]; double d[2][3];
            for (int u = 0; u < 2; u++) particle(cf, (it * 1024 + u * 512 + (int)threadIdx.x) & 4095, it, lb[u], d[u]);
            if (cf.swap && (lane & 1)) for (int c = 0; c < 3; c++) { int t = lb[0][c]; lb[0][c] = lb[1][c]; lb[1][c] = t; double td = d[0][c]; d[0][c] = d[1][c]; d[1][c] = td; }
            for (int u = 0; u < 2; u++) {
                double V[3][4];
                for (int c = 0; c < 3; c++) pcs(d[u][c], V[c]);
                for (int a = 0; a < 4; a++) V[0][a] *= ms;
                const int base = (lb[u][0] * R1 + lb[u][1]) * PITCH + lb[u][2];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const double f = V[0][a] * V[1][b];
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const double r = __builtin_fma(f, V[2][c], MAGIC);
                            if (cf.noatom) sink += r;
                            else atomicAdd((unsigned long long *)&lds[base + (a * R1 + b) * PITCH + c], (unsigned long long)(__double_as_longlong(r) - MAGIC_BITS));
                        }
                    }
            }
        }
    } else {
        const int q = lane & 3;
        // the cubic of stencil index q in the offset d (PCS): w0 = (1-d)^3/6, w1 = 2/3 - d^2 + d^3/2, w2 = 2/3 - (1-d)^2 + (1-d)^3/2 ... as coefficients
        const double C[4][4] = {{1.0 / 6, -0.5, 0.5, -1.0 / 6}, {2.0 / 3, 0.0, -1.0, 0.5}, {1.0 / 6, 0.5, 0.5, -0.5}, {0.0, 0.0, 0.0, 1.0 / 6}};
        double co[4];
        for (int i = 0; i < 4; i++) co[i] = C[q][i];
        for (int it = 0; it < iters; it++) {
            for (int u = 0; u < 2; u++) {
                int lb[3]; double d[3];
                particle(cf, (it * 1024 + u * 512 + (int)threadIdx.x) & 4095, it, lb, d);
                const int base = (lb[0] * R1 + lb[1]) * PITCH + lb[2];
                double dd[3];
#define SUB(CT) dd[0] = quad_bcast<CT>(d[0]); dd[1] = quad_bcast<CT>(d[1]); dd[2] = quad_bcast<CT>(d[2]); \
                quad_subtrip<PITCH, AX>(cf, lds, dd, quad_bcast<CT>(base), q, co, ms, sink);
                SUB(0x00) SUB(0x55) SUB(0xaa) SUB(0xff)
#undef SUB
            }
        }
    }
    __syncthreads();
    double s = sink;
    for (int i = threadIdx.x; i < cells; i += 512) s += lds[i];
    if (s == 12345.0) out[blockIdx.x] = s;
}

template <int PITCH, int AX>
static void run(Cfg cf, const char *name)
{
    double *out; (void)hipMalloc(&out, 1 << 20);
    const size_t lds = (size_t)R0 * R1 * PITCH * 8;
    int wgs = (int)(160 * 1024 / (lds + 512)); if (wgs > 4) wgs = 4;
    const int blocks = 256 * wgs * 4, iters = 40;
    auto kern = k<PITCH, AX>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    kern<<<blocks, 512, lds>>>(cf, out, 3);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    kern<<<blocks, 512, lds>>>(cf, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double particles = (double)blocks * iters * 1024;
    const char *orders[] = {"perfect lattice", "jittered lattice", "random in the tile", "clustered (half in 27 cells)"};
    printf("%-44s pitch %2d %-28s %s LDS %4.1f KB x %d: %6.2f clk per particle and CU\n", name, PITCH, orders[cf.order], cf.noatom ? "NO ATOMICS" : "          ",
           lds / 1024.0, wgs, (ms * 1e-3) * 2.4e9 * 256 / particles);
    (void)hipFree(out);
}
int main()
{
    for (int na = 0; na < 2; na++)
        for (int order = 0; order < 4; order++) {
            run<48, 0>(Cfg{0, order, 1, na}, "lane per particle, swap (today)");
            if (!na) run<48, 0>(Cfg{0, order, 0, na}, "lane per particle, no swap");
            if (!na) run<36, 0>(Cfg{0, order, 1, na}, "lane per particle, swap");
            run<48, 0>(Cfg{1, order, 0, na}, "quad along x");
            run<36, 0>(Cfg{1, order, 0, na}, "quad along x");
            run<35, 0>(Cfg{1, order, 0, na}, "quad along x");
            run<48, 1>(Cfg{2, order, 0, na}, "quad along y");
            run<36, 1>(Cfg{2, order, 0, na}, "quad along y");
            run<40, 1>(Cfg{2, order, 0, na}, "quad along y");
            run<48, 2>(Cfg{3, order, 0, na}, "quad along z");
            run<36, 2>(Cfg{3, order, 0, na}, "quad along z");
            run<35, 2>(Cfg{3, order, 0, na}, "quad along z");
        }
    return 0;
}
