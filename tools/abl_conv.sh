#!/bin/bash
# A/B timing of the conv layers with variant libraries in detectorfreesfm_amd/csrc/abl/ (built with -D DFSFM_ABL_<V>)
MODE=${1:-same}; shift
for v in "" "$@"; do
  if [ -z "$v" ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so; fi
  echo "== variant: ${v:-full}"
  timeout 100 python tools/bench_conv_layers.py $MODE 2>&1 | grep -E "${FILTER:-3x3|conv|adap|1x1}"
done
