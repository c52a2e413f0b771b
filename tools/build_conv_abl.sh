#!/bin/bash
# Timing-only ablation builds of the "same" convolution main loop (csrc/sf_gemm.h; results are wrong): csrc/abl/lib_conv_<switch>.so
set -e
cd "$(dirname "$0")/../detectorfreesfm_amd/csrc"
mkdir -p abl build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-function -fno-slp-vectorize -I../../include -I."
OTHERS=$(ls build/*.o | grep -v conv_gemm)
for v in "$@"; do
  defs="-DSF_${v//+/ -DSF_}"                 # ABL_NOMFMA+ABL_NODMA -> -DSF_ABL_NOMFMA -DSF_ABL_NODMA
  /opt/rocm/bin/hipcc $FLAGS $defs -c conv_gemm.hip -o abl/conv_gemm_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map $OTHERS abl/conv_gemm_$v.o -o abl/lib_conv_$v.so
  echo built abl/lib_conv_$v.so
done
