#!/bin/bash
# Final measurement session of the second r05 session on the frozen sources: the PMC traffic file, then the default bench line
# (its library_source_sha256 must match the sources bench.py times: traffic_build_matches).
exec < /dev/null
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/final; export TMPDIR=/tmp
timeout 230 python tools/pmc_collect.py gpurun_out/final/r05_pmc_traffic.json --scratch gpurun_out/final/pmc_scratch > gpurun_out/final/pmc.log 2>&1; tail -1 gpurun_out/final/pmc.log; rm -rf gpurun_out/final/pmc_scratch
python -c "import json; d=json.load(open('gpurun_out/final/r05_pmc_traffic.json')); assert len(d['kernels']) > 10" && cp gpurun_out/final/r05_pmc_traffic.json profiles/r05_pmc_traffic.json
timeout 200 python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"; python - <<'P'
import json
d = json.loads(open("gpurun_out/final/bench_n1.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "traffic_build_matches") if k in d}, d.get("secondary", {}).get("value"), d.get("feeding"), d["roofline"]["frac"])
P
