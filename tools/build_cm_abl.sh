#!/bin/bash
# Timing-only ablation builds of csrc/coarse_match.hip's panel kernel (results are wrong): csrc/abl/lib_cm_<switch>.so
set -e
cd "$(dirname "$0")/../detectorfreesfm_amd/csrc"
mkdir -p abl build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-function -fno-slp-vectorize -I../../include -I."
OTHERS=$(ls build/*.o | grep -v coarse_match)
for v in ${@:-NOEPI NOLOOP}; do   # NOEPI NOLOOP VMSLACK
  /opt/rocm/bin/hipcc $FLAGS -DCM_PANEL_$v -c coarse_match.hip -o abl/coarse_match_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map $OTHERS abl/coarse_match_$v.o -o abl/lib_cm_$v.so
  echo built abl/lib_cm_$v.so
done
