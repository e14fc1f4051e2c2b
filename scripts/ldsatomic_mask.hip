// What does a ds_add_f64 cost when only some lanes are active?  (Deciding whether merging the deposits of
// neighbouring lanes in registers — fewer lane-atomics, same number of instructions — can pay.)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/ldsatomic_mask.hip -o scripts/ldsatomic_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int CELLS = 6144;
// MODE 0: ds_add_f64, 1: ds_add_u64, 2: ds_add_u32, 3: ds_write_b64, 4: ds_add_f64 on random cells
template <int MODE>
__global__ void __launch_bounds__(512) k(double *out, int iters, unsigned long long mask)
{
    __shared__ double lds[CELLS];
    for (int q = threadIdx.x; q < CELLS; q += 512) lds[q] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool active = (mask >> lane) & 1;
    double v = (double)(threadIdx.x + 1);
    uint32_t h = threadIdx.x * 2654435761u;
    for (int it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        int base = MODE == 4 ? (int)((h >> 8) % (CELLS - 64 * 27)) : ((wave * 2 + it) & 31) * 64 + lane;
        if (active) {
#pragma unroll
            for (int c = 0; c < 27; c++) {
                if (MODE == 0 || MODE == 4) unsafeAtomicAdd(&lds[base + c * 64], v);
                else if (MODE == 1) atomicAdd((unsigned long long *)&lds[base + c * 64], (unsigned long long)threadIdx.x);
                else if (MODE == 2) atomicAdd((unsigned int *)&lds[base + c * 64], (unsigned int)threadIdx.x);
                else *(volatile double *)&lds[base + c * 64] = v;
            }
        }
    }
    __syncthreads();
    double s = 0;
    for (int q = threadIdx.x; q < CELLS; q += 512) s += lds[q];
    if (s == 12345.0) out[blockIdx.x] = s;
}
template <int MODE> void run(const char *name, unsigned long long mask)
{
    double *out; hipMalloc(&out, 1 << 20);
    int blocks = 256 * 3 * 4, iters = 200;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 512>>>(out, 10, mask);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 512>>>(out, iters, mask);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr = (double)blocks * 8 * iters * 27;            // wave instructions
    printf("%-28s lanes %2d: %.3f ms: %.1f clk per wave-instruction per CU\n", name, __builtin_popcountll(mask), ms,
           (ms * 1e-3) * 2.4e9 * 256 / instr);
    hipFree(out);
}
int main()
{
    const unsigned long long masks[] = {~0ull, 0x5555555555555555ull, 0x1111111111111111ull, 0x0101010101010101ull,
                                        0x00000000FFFFFFFFull, 0x000000000000FFFFull, 0x000000000000000Full, 1ull};
    for (auto m : masks) run<0>("ds_add_f64 lane-linear", m);
    for (auto m : masks) run<1>("ds_add_u64 lane-linear", m);
    for (auto m : masks) run<2>("ds_add_u32 lane-linear", m);
    for (auto m : masks) run<3>("ds_write_b64 lane-linear", m);
    for (auto m : masks) run<4>("ds_add_f64 random cells", m);
    return 0;
}
