#!/bin/bash
# r06 session ad: encoder256.hip with MFMA results in the vector file -- the kv kernel by itself (bench.py --kernels-only rows)
exec < /dev/null
export TMPDIR=/tmp
for r in 1 2; do
for v in prod e256_vf; do
  if [ $v = prod ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so; fi
  echo "== $v"
  timeout 600 python bench.py --kernels-only 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['rooflines']:
    if 'enc256' in r['kernel'] or 'coarse encoder' in r['kernel']: print('  %-100s %.4f ms' % (r['kernel'][:100], r['ms']))
"
done
done
