#!/bin/bash
# r06 measurement session on the frozen sources: default bench line, workloads, step / kernel stats, PMC traffic
exec < /dev/null
tag=${1:-r6z}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_measure.sh $tag quick
timeout 900 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_n1.json"))
print({k:d.get(k) for k in ("value","ms_per_step","refinement_tracks_per_sec","refinement_ms_per_step","traffic_build_matches")}, d["roofline"]["frac"], d["pipelined"]["value"])
for r in d["rooflines"]: print(r["kernel"][:60], round(r["frac"],3), r.get("traffic"))
PY
