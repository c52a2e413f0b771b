#!/bin/bash
# Variant libraries for A/B ablations: detectorfreesfm_amd/csrc/abl/lib_<V>.so = the product library with conv_gemm.hip
# compiled with -DDFSFM_ABL_<V> (all other objects shared).  usage: tools/build_abl.sh NOEPI OOBA ...
set -e
C=detectorfreesfm_amd/csrc
make -C $C -j8 > /dev/null
mkdir -p $C/abl
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-function -Iinclude -I$C -DDFSFM_ABL_$v -c $C/conv_gemm.hip -o $C/abl/conv_gemm_$v.o
  objs=$(ls $C/build/*.o | grep -v conv_gemm.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map $objs $C/abl/conv_gemm_$v.o -o $C/abl/lib_$v.so
  echo built $C/abl/lib_$v.so
done
