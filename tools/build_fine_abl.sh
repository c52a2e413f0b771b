#!/bin/bash
# Experiment builds of csrc/fine_match.hip for the two-workgroups-per-CU nondeterminism: csrc/abl/lib_fine_<v>.so
set -e
cd "$(dirname "$0")/../detectorfreesfm_amd/csrc"
mkdir -p abl build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-function -I../../include -I."
OTHERS=$(ls build/*.o | grep -v fine_match)
build() { name=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c fine_match.hip -o abl/fine_match_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map $OTHERS abl/fine_match_$name.o -o abl/lib_fine_$name.so
  echo built abl/lib_fine_$name.so; }
# the shipped schedule is two workgroups per CU, built with -fno-slp-vectorize (Makefile); -DFINE_1WG = the r03 schedule
build 2WG_PACKED                                   # the bug: two workgroups per CU WITH packed-fp32 instructions
build 2WG_PACKED_NOTAIL -DFINE_NO_OOB_TAIL
build 2WG_PACKED_DRAIN -DFINE_DRAIN
build 2WG_PACKED_BARRIER -DFINE_BARRIER
build 2WG_PACKED_DRAIN_BARRIER -DFINE_DRAIN -DFINE_BARRIER
build 2WG_PACKED_SLEEP -DFINE_SLEEP
build 2WG_PACKED_SLACK -DFINE_SLACK
build 2WG_PACKED_NODMA -DFINE_NODMA
build 2WG_PACKED_SETTLE -DFINE_MFMA_SETTLE
build 2WG_NOPK -Xclang -target-feature -Xclang -packed-fp32-ops
build 2WG_NOSLP -fno-slp-vectorize                 # = the product build of this file
build 1WG_PACKED -DFINE_1WG                        # r03
build 1WG_NOSLP -DFINE_1WG -fno-slp-vectorize
