# SQ wave-cycle counters and dynamic instruction counts of the particle kernels (round 5's forms): headline, config 3, PCS
for cfg in "headline:" "c3:--window tsc --dtype f4 --gradient 0" "pcs:--window pcs" "tsc:--window tsc"; do
  n=${cfg%%:*}; a=${cfg#*:}
  scripts/sq_profile.sh r05_sq_$n $a > /dev/null 2>&1
  scripts/inst_profile.sh r05_inst_$n $a > /dev/null 2>&1
  echo "== $n ($a)"; cat gpurun_out/r05_sq_$n/sq_summary.txt | grep -v "fft\|halo"; cat gpurun_out/r05_inst_$n/inst_summary.txt
done 2>&1 | cut -c1-330 | tee gpurun_out/r05_sq_inst.txt
