#!/bin/bash
# r05 FINAL session on the frozen sources: full GPU suite, smoke, kernels-only stats, PMC traffic (copied into profiles/ on the box so the
# bench line of the same session can state traffic_build_matches), default bench with cpu_baseline, warm step profiles, explicit workloads
exec < /dev/null
out=gpurun_out/r5z; mkdir -p $out
export TMPDIR=/tmp
root=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_k -o k -- python bench.py --kernels-only > $out/kernels_only.json 2> $out/prof_k.log
f=$(find $out/prof_k -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernels_only_kernel_stats.csv
rm -rf $out/prof_k
timeout 900 python tools/pmc_collect.py $out/pmc_traffic.json --scratch $out/pmc_s > $out/pmc.log 2>&1; rm -rf $out/pmc_s
cp $out/pmc_traffic.json profiles/r05_pmc_traffic.json
timeout 900 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
cd /tmp
for w in coarse refine; do
  timeout 600 env PYTHONPATH=$root rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$w -o $w -- python $root/tools/profile_step.py $w 10 > $root/$out/prof_$w.log 2>&1
  f=$(find $root/$out/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/$out/${w}_step_kernel_stats.csv
  rm -rf $root/$out/prof_$w
done
cd $root
timeout 400 python bench.py --workload scene300 --no-cpu-baseline > $out/scene300.json 2> $out/scene300.err
timeout 300 python bench.py --workload scene300 --scene-images 60 > $out/scene60.json 2> $out/scene60.err
DFSFM_BENCH_FORCE_DIST=1 timeout 300 python bench.py --workload scene300 --scene-images 60 > $out/scene60_dist.json 2> $out/scene60_dist.err
timeout 400 python bench.py --workload hires832 --steps 6 --warmup 2 --no-cpu-baseline > $out/hires832.json 2> $out/hires832.err
timeout 300 python bench.py --workload eth3d1600 --steps 5 --warmup 2 --no-cpu-baseline > $out/eth3d1600.json 2> $out/eth3d1600.err
timeout 300 python bench.py --workload demo1200 --steps 5 --warmup 2 --no-cpu-baseline > $out/demo1200.json 2> $out/demo1200.err
timeout 300 python bench.py --workload matchformer --steps 6 --warmup 2 --no-cpu-baseline > $out/matchformer.json 2> $out/mf.err
timeout 300 python bench.py --workload aspanformer --steps 6 --warmup 2 --no-cpu-baseline > $out/aspanformer.json 2> $out/as.err
timeout 300 python bench.py --workload scene300 --scene-matcher aspanformer --scene-images 40 > $out/scene40_aspanformer.json 2> $out/scene40_as.err
timeout 300 python tools/bench_cm.py --big --split-only 2>&1 | grep "ms / call" > $out/bench_cm.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5z/bench_n1.json'))
print({k:d.get(k) for k in ('value','ms_per_step','breakdown','traffic_build_matches')}, d['pipelined']['value'], d['secondary']['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for f in ('scene300','scene60','hires832','eth3d1600','demo1200','matchformer','aspanformer','scene40_aspanformer'):
    try:
        e=json.load(open(f'gpurun_out/r5z/{f}.json')); print(f, round(e['value'],1), e.get('breakdown'))
    except Exception as ex: print(f,'ERR',ex)
PY
