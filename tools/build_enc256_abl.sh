#!/bin/bash
# Timing-only ablation builds of csrc/encoder256.hip (results are wrong with any switch set): one library per switch under
# detectorfreesfm_amd/csrc/abl/ (git-ignored; travels with gpurun).  tools/bench_enc256.py times them through DFSFM_LIB_PATH.
set -e
cd "$(dirname "$0")/../detectorfreesfm_amd/csrc"
mkdir -p abl build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-function -I../../include -I."
OTHERS=$(ls build/*.o | grep -v encoder256)
for v in NOMFMA NODMA NOBAR NOREAD "NODMA -DENC256_NOBAR" "NODMA -DENC256_NOBAR -DENC256_NOREAD"; do
  name=$(echo "$v" | sed 's/ -DENC256_/_/g')
  /opt/rocm/bin/hipcc $FLAGS -DENC256_$v -c encoder256.hip -o abl/encoder256_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map $OTHERS abl/encoder256_$name.o -o abl/lib_enc256_$name.so
  echo built abl/lib_enc256_$name.so
done
