#!/bin/bash
# Image-feeding session: GPU tests of dfsfm_resample_u8 / images.py, frames/s next to Pillow on the host.
exec < /dev/null
out=gpurun_out/${1:-img1}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_images.py -q > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
timeout 300 python tools/bench_images.py > $out/bench.log 2>&1
tail -25 $out/pytest.log | cut -c1-250; tail -3 $out/bench.log
