#!/bin/bash
# r06 session ae: the 7x7 stem on its own kernel (since dropped: tools/studies/stem_conv_r06_not_kept.hip.txt, profiles/r06_encoder_stream_diag.txt section 5) -- kept as the record of what was run
exec < /dev/null
tag=${1:-r6ae}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "stem7x7s2" > $out/pytest_stem.log 2>&1; tail -4 $out/pytest_stem.log
python - > $out/stem_time.txt 2>&1 <<'PY'
import torch, time
from detectorfreesfm_amd import ops
DEV="cuda:0"
g=torch.Generator().manual_seed(0)
w=torch.randn(128,1,7,7,generator=g)*0.2; b=torch.randn(128,generator=g)
x=torch.rand(16,480,640,1,generator=g).to(DEV)
sw=ops.StemWeights(w.to(DEV),b.to(DEV)); pw=ops.PackedDense(w.to(DEV),b.to(DEV))
def ev(fn,it=20):
    for _ in range(3): fn()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize(); return s.elapsed_time(e)/it
t1=ev(lambda: ops.stem7x7s2(x,sw)); t0=ev(lambda: ops.conv2d_nhwc(x,pw,2,3,relu=True,out_split=True))
print(f"stem 16 x 480x640: own kernel {t1:.3f} ms, implicit GEMM {t0:.3f} ms; bytes written 629 MB -> {0.629/t1:.2f} TB/s")
PY
cat $out/stem_time.txt | grep -v amdgpu.ids
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -x -k "loftr or coarse" > $out/pytest_e2e.log 2>&1; tail -4 $out/pytest_e2e.log
for r in 1 2; do
for f in True False; do
python -c "
import sys
import detectorfreesfm_amd.coarse as c
c.STEM_KERNEL = $f
sys.argv=['bench.py','--steps','10','--warmup','2','--no-cpu-baseline','--no-rooflines']
import bench
bench.main()
" 2>/dev/null | grep -o "\"value\": [0-9.]*\|\"backbone_ms\": [0-9.]*" | head -3 | tr "\n" " "; echo " STEM_KERNEL=$f"
done
done
