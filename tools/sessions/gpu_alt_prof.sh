#!/bin/bash
# rocprofv3 kernel stats of the two alternative coarse matchers (one warm-up forward + n-1 timed forwards).
exec < /dev/null
tag=${1:-alt1}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
root=$PWD
cd /tmp
for w in ${WHICH:-aspan matchformer}; do
  timeout 300 env PYTHONPATH=$root python $root/tools/profile_step.py $w 9 > $root/$out/wall_$w.log 2>&1
  timeout 600 env PYTHONPATH=$root rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$w -o $w -- python $root/tools/profile_step.py $w 9 > $root/$out/prof_$w.log 2>&1
  f=$(find $root/$out/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/$out/${w}_kernel_stats.csv
  rm -rf $root/$out/prof_$w
  cat $root/$out/wall_$w.log | tail -1
  python - <<PY
import csv
rows=list(csv.DictReader(open("$root/$out/${w}_kernel_stats.csv")))
tot=sum(int(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print("$w: kernel time %.3f ms per forward, %d launches per forward" % (tot/9e6, calls/9))
for r in rows[:12]: print('  ', r['Name'][:80], r['Calls'], '%.3f ms' % (int(r['TotalDurationNs'])/9e6), r['Percentage'])
PY
done
