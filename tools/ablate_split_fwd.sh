#!/bin/bash
# Developer tool: timing-only ablations of the split-operand k-loop of
# fwd_head_loss_split_kernel (results are WRONG in these builds); run
# GARAGE_AMD_SPLIT_BF16=1 GA_VARIANT_LIB=garage_amd/_C/variants/lib_<name>.so
#   python tools/fused_fwd_phases.py c3 64
set -e
cd "$(dirname "$0")/.."
make -s
OUT=garage_amd/_C/variants
mkdir -p $OUT
build() {  # name, flags...
  name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function "$@" -c garage_amd/csrc/fused_train.hip -o $OUT/ft_$name.o
  objs=$(ls garage_amd/_C/*.o | grep -v '/fused_train.o')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $OUT/ft_$name.o -ldl -o $OUT/lib_$name.so
  echo built $name
}
build noproduce -DGA_ABL_NOPRODUCE &
build nobarrier -DGA_ABL_NOBARRIER &
build nofetch -DGA_ABL_NOFETCH &
build nomfma -DGA_ABL_NOMFMA &
wait
build nothing -DGA_ABL_NOPRODUCE -DGA_ABL_NOFETCH &
build bare -DGA_ABL_NOPRODUCE -DGA_ABL_NOFETCH -DGA_ABL_NOBARRIER &
wait
