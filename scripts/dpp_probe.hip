// which cross-lane shifts does gfx950 really do?   hipcc --offload-arch=gfx950 -O2 scripts/dpp_probe.hip -o scripts/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x138, 0xf, 0xf, false);            // wave_shr:1
    out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x130, 0xf, 0xf, false);       // wave_shl:1
    int t = __builtin_amdgcn_update_dpp(-1, lane, 0x142, 0xf, 0xf, false);                // row_bcast:15
    out[128 + lane] = t;
    out[192 + lane] = __builtin_amdgcn_update_dpp(t, lane, 0x111, 0xf, 0xf, false);       // row_shr:1 over it
    out[256 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x101, 0xf, 0xf, false);      // row_shl:1
    out[320 + lane] = __builtin_amdgcn_ds_bpermute((lane - 1) << 2, lane);
    out[384 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x13C, 0xf, 0xf, false);      // wave_ror:1
}
int main()
{
    int *d, h[448];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"wave_shr:1", "wave_shl:1", "row_bcast:15", "row_shr:1 over row_bcast:15", "row_shl:1", "ds_bpermute lane-1", "wave_ror:1"};
    for (int r = 0; r < 7; r++) {
        printf("%-28s", names[r]);
        for (int l = 0; l < 64; l++) printf(" %d", h[r * 64 + l]);
        printf("\n");
    }
    return 0;
}
