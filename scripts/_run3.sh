for of in 1 0; do
  echo "== out_field $of L3=0"; PMESH_AMD_L3_BLOCK_MB=0 bash scripts/kstats.sh ks_$of --out-field $of 2>&1 | grep -i "rowfft\|colfft\|halo\|paint_tile" 
  echo "== out_field $of L3=224"; bash scripts/kstats.sh ks224_$of --out-field $of 2>&1 | grep -i "rowfft\|colfft\|halo\|paint_tile" 
done
