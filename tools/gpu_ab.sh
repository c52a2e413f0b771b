#!/bin/bash
# Same-box A/B of two builds of the library: A = detectorfreesfm_amd/csrc/abl/lib_<name>.so, B = the product build.
# usage: tools/gpu_ab.sh <name> [rounds]   (box-to-box variance is ~4 %, same-box repeatability ~0.3 %)
exec < /dev/null
name=${1:-base}; rounds=${2:-2}
for r in $(seq $rounds); do
  for v in A B; do
    if [ $v = A ]; then export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$name.so; else unset DFSFM_LIB_PATH; fi
    echo -n "== $v "
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rooflines 2>/dev/null | grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" | head -4 | tr "\n" " "
    echo
  done
done
