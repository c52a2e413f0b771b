#!/bin/bash
# Same-box timing of the backbone layer shapes with several builds of the library (csrc/abl/lib_<name>.so; "product" = the shipped one)
exec < /dev/null
for name in "$@"; do
  if [ "$name" = product ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$name.so; fi
  echo "== $name"
  python tools/bench_conv_layers.py same 2>&1 | sed -n "3,5p;7,8p"
done
