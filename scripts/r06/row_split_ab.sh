#!/bin/bash
# A/B of the pencil transform's last-axis split on the row pass (pmx_rowfft_split, fft.ROW_SPLIT) against the row pass +
# pmx_slab_pack: 8 thread ranks on one GPU, 2 x 4 pencils, per-kernel sums.   scripts/r06/row_split_ab.sh <tag> [mesh] [steps]
tag=${1:-r06_rowsplit}
mesh=${2:-512}
steps=${3:-12}
mkdir -p gpurun_out/$tag
for split in 0 1; do
    export PMESH_AMD_ROW_SPLIT=$split
    scripts/mr_kstats.sh ${tag}_${mesh}_split${split} --ranks 8 --mesh $mesh --steps $steps --warmup 2 --np 2x4 --migrate 1 2>&1 | grep "wall per cycle\|pmx kernels\|rowfft\|slab_pack"
done
