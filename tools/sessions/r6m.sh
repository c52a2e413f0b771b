#!/bin/bash
# r06 checkpoint: whole GPU suite, smoke, bench (with cpu_baseline)
exec < /dev/null
tag=${1:-r6m}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "passed|failed|error" $out/pytest.log | tail -5; grep -E "^FAILED|^ERROR" $out/pytest.log | head
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log; tail -2 $out/smoke.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
python - <<PY
import json
d=json.load(open("$out/bench.json"))
print({k:d.get(k) for k in ("value","ms_per_step","refinement_tracks_per_sec","refinement_ms_per_step")})
print(d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("real_module_ratio"))
print(d["feeding"])
print([ (r["kernel"][:40], round(r["frac"],3), r.get("traffic")) for r in d["rooflines"] if "fine_match" in r["kernel"]])
PY
tail -2 $out/bench.err
