"""bench.py's driver contract, checked without a GPU: the recorded line of the final run carries every field the contract
names (plus `roofline` and `cpu_baseline`), the command line parses, `--gpus N` spawns N ranks on 127.0.0.1, and a machine
without a GPU gets a loud refusal, not a CPU number."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_recorded_line_has_the_contract_fields():
    line = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_final_n1.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert "configs[1]" in line["config"]["workload"] and line["unit"] == "image-pairs/s"
    assert abs(line["value"] - 8 * 1000.0 / line["ms_per_step"]) < 1e-6 * line["value"]           # pairs per step / time per step
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert line["secondary"]["metric"] == "refinement_tracks_per_sec" and line["secondary"]["value"] > 0
    assert line["matches_last_step"] > 0                                                           # the timed step produces tables


def test_cli_defaults_and_workloads():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--kernels-only"):
        assert flag in out.stdout
    for w in ("pairs", "scene300", "hires832", "matchformer", "aspanformer"):
        assert w in out.stdout


def test_self_spawn_command(monkeypatch):
    b = _bench()
    seen = {}
    monkeypatch.setattr(b.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert b._self_spawn(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_no_gpu_no_number():
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode != 0 and "no CPU fallback" in (out.stderr + out.stdout)
    assert not any(l.strip().startswith("{") for l in out.stdout.splitlines())
