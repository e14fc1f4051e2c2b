#!/bin/bash
# scripts/build_variant.sh <name> "<EXTRA flags>" [file.hip ...]: builds pmesh_amd/libpmesh_amd_<name>.so with the
# given files recompiled under EXTRA (default: pmx_binned.hip), the other objects of the product build as they are.
# (pmx_binned.hip and pmx_colfft.hip are built in parts by the Makefile; a variant compiles the whole file as one unit.)
name=$1; flags=$2; shift 2
files=${@:-pmx_binned.hip}
cd pmesh_amd/csrc
objs=""
for f in pmx_core pmx_window pmx_binned pmx_domain pmx_transfer pmx_synth pmx_fft pmx_colfft pmx_whitenoise; do
  if [ $f = pmx_core ]; then
    # the variant says what it is: pmx_build_flags() = the product's compiler line + the files and flags of the variant
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -I../../include "-DPMX_VARIANT_FLAGS=\" | variant $name: $files: $flags\"" -c $f.hip -o /tmp/${f}_$name.o || exit 1
    objs="$objs /tmp/${f}_$name.o"
  elif echo " $files " | grep -q " $f.hip "; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -I../../include $flags -c $f.hip -o /tmp/${f}_$name.o || exit 1
    objs="$objs /tmp/${f}_$name.o"
  else
    objs="$objs $f.o"
    [ $f = pmx_binned ] && objs="$objs pmx_binned_paint.o pmx_binned_paint_f4.o pmx_binned_readout.o"
    [ $f = pmx_colfft ] && objs="$objs pmx_colfft_f4.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpmesh_amd_$name.so $objs -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
