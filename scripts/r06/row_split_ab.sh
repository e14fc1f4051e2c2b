#!/bin/bash
# A/B of the pencil transform's last-axis split on the row pass (pmx_rowfft_split, fft.ROW_SPLIT) against the row pass +
# pmx_slab_pack: 8 thread ranks on one GPU, 2 x 4 pencils at 512^3, per-kernel sums.   scripts/r06/row_split_ab.sh <tag>
tag=${1:-r06_rowsplit}
mkdir -p gpurun_out/$tag
for split in 0 1; do
    export PMESH_AMD_ROW_SPLIT=$split
    scripts/mr_kstats.sh ${tag}_split${split} --ranks 8 --mesh 512 --steps 12 --warmup 2 --np 2x4 --migrate 1 2>&1 | tail -4
done
