#!/bin/bash
# Developer tool: builds libgarage_amd variants with different GEMM tile macros
# into garage_amd/_C/variants/ for A/B runs (tools/gemm_sweep.py --lib ...).
set -e
cd "$(dirname "$0")/.."
make -s
OUT=garage_amd/_C/variants
mkdir -p $OUT
build() {  # name, flags...
  name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c garage_amd/csrc/gemm.hip -o $OUT/gemm_$name.o
  objs=$(ls garage_amd/_C/*.o | grep -v '/gemm.o')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $OUT/gemm_$name.o -ldl -o $OUT/lib_$name.so
  echo built $name
}
build occ6 -DGA_GEMM_WAVES_PER_EU=6 &
wait
