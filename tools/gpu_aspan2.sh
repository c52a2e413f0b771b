#!/bin/bash
# ASpanFormer: GPU tests, timing, bench.py lines of the two alternative matchers.
exec < /dev/null
out=gpurun_out/${1:-asb}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_aspan.py -q -s 2>&1 | grep -v "^  File\|pluggy\|_pytest" | tail -30 > $out/pytest_all.log
timeout 300 python tools/bench_aspanformer.py > $out/bench.log 2>&1
timeout 300 python bench.py --workload aspanformer --steps 6 --warmup 2 > $out/bench_aspanformer.json 2> $out/bench_aspanformer.err
timeout 300 python bench.py --workload matchformer --steps 6 --warmup 2 > $out/bench_matchformer.json 2> $out/bench_matchformer.err
grep -n "passed\|failed\|^\[aspan\|Error\|assert \|fault" $out/pytest_all.log | cut -c1-300 | head -12
tail -1 $out/bench.log; cut -c1-330 $out/bench_aspanformer.json; cut -c1-330 $out/bench_matchformer.json; tail -2 $out/bench_matchformer.err
