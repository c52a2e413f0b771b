#!/bin/bash
# Which kernels of the library mix MFMA and packed-fp32 VALU instructions, and at what occupancy?  (r04: packed fp32 beside another
# wave's MFMAs on the same SIMD gave wrong results in fine_match; a kernel is exposed when BOTH appear and more than one wave per
# SIMD can be resident.)  usage: tools/scan_packed_fp32.sh > profiles/rNN_packed_fp32_scan.txt
cd "$(dirname "$0")/../detectorfreesfm_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-function -I../../include -I."
printf "%-70s %6s %6s %6s %s\n" kernel mfma pk_f32 occ file
for f in *.hip; do
  extra=""; [ "$f" = fine_match.hip ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc $FLAGS $extra -S --cuda-device-only $f -o /tmp/scan.s -Rpass-analysis=kernel-resource-usage 2> /tmp/scan.rem
  python3 - "$f" <<'PY'
import re, sys
f = sys.argv[1]
asm = open('/tmp/scan.s').read()
rem = open('/tmp/scan.rem').read()
occ = {}
cur = None
for line in rem.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m: cur = m.group(1)
    m = re.search(r'Occupancy \[waves/SIMD\]: (\d+)', line)
    if m and cur: occ[cur] = int(m.group(1))
for name, body in re.findall(r'^(_Z\w+):[^\n]*\n(.*?)^\s*\.amdhsa_kernel', asm, re.S | re.M):
    mf = len(re.findall(r'v_mfma', body)); pk = len(re.findall(r'v_pk_(?:add|mul|fma)_f32', body))
    if mf and pk:
        import subprocess
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:70]
        print(f"{dem:70s} {mf:6d} {pk:6d} {occ.get(name, '?'):>6} {f}")
PY
done
