#!/bin/bash
# r06 closing session on the frozen sources: PMC traffic, the default bench line (reads the PMC file of the same build), its rocprof stats, the GPU suite, smoke
exec < /dev/null
tag=${1:-r6zs}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
root=$PWD
timeout 900 python tools/pmc_collect.py $out/pmc_traffic.json --scratch $out/pmc_s > $out/pmc.log 2>&1; rm -rf $out/pmc_s
[ -s $out/pmc_traffic.json ] && cp $out/pmc_traffic.json profiles/r06_pmc_traffic.json
timeout 900 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_n1.json"))
print({k:d.get(k) for k in ("value","ms_per_step","refinement_tracks_per_sec","refinement_ms_per_step","traffic_build_matches")}, d["roofline"]["frac"])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_b -o b -- python $root/bench.py --no-cpu-baseline > $root/$out/bench_profiled.json 2> $root/$out/prof_b.log)
f=$(find $out/prof_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/final_bench_kernel_stats.csv
rm -rf $out/prof_b
timeout 1500 python -m pytest tests -m gpu -x -q > $out/final_gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $out/final_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
