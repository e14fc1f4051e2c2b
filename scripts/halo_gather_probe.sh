#!/bin/bash
# scripts/halo_gather_probe.sh: the halo merge as a kernel of its own (PMESH_AMD_HALO_DEFER=never) against the gather in the
# forward row pass (fresh), kernel by kernel (rocprofv3 averages over 8 + 2 cycles) -> gpurun_out/halo_gather_probe.txt
out=gpurun_out/halo_gather_probe.txt; : > $out
for cfg in "headline:" "c3:--window tsc --dtype f4 --gradient 0" "tsc_f8:--window tsc" "pcs_f8:--window pcs" "cic_f4:--dtype f4" "m256:--mesh 256" "m1024:--mesh 1024 --steps 3"; do
  n=${cfg%%:*}; a=${cfg#*:}
  for mode in never fresh; do
    echo "== $n ($a) PMESH_AMD_HALO_DEFER=$mode" >> $out
    PMESH_AMD_HALO_DEFER=$mode bash scripts/kstats.sh hg_${n}_$mode $a 2>&1 | grep "paint_tile_kernel\|halo_merge_kernel\|rowfft_kernel<[a-z]*, [0-9]*, false" | sed 's/(pmx.*calls/ calls/; s/(pmx_painter.*calls/ calls/' | cut -c1-200 >> $out
  done
done
cat $out
