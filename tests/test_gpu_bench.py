"""bench.py keeps its contract: one JSON line, last on stdout, with the fields the driver reads; the multi-GPU code paths run (in a
one-rank group, MPG_FORCE_MGPU) and report the same kind of line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline"}


def run_bench(args, env=None):
    e = dict(os.environ, **(env or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    return json.loads(last)


def test_default_workload_line():
    j = run_bench(["--size", "64", "--steps", "2", "--warmup", "1", "--cpu-sample", "65536"])
    assert KEYS <= set(j) and "cpu_baseline" in j
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["dtype"] == "f64"
    assert j["value"] > 1e6 and abs(j["value"] - 64 ** 3 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    assert set(j["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and j["cpu_baseline"]["kind"] == "port"
    assert "workload" in j["config"]


@pytest.mark.parametrize("mode", ["domain", "slab", "replicated"])
def test_multi_gpu_paths_in_a_one_rank_group(mode):
    j = run_bench(["--gpus", "1", "--size", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--mgpu", mode],
                  env={"MPG_FORCE_MGPU": "1", "MASTER_PORT": "29%03d" % (hash(mode) % 900 + 50)})
    assert KEYS <= set(j) and j["value"] > 1e6


def test_other_workloads():
    h = run_bench(["--workload", "hydro", "--size", "32", "--steps", "1", "--warmup", "0"])
    assert KEYS <= set(h) and "roofline_hydro" in h
    i = run_bench(["--workload", "integrate", "--size", "64", "--steps", "2", "--warmup", "1"])
    assert KEYS <= set(i) and i["roofline"]["frac"] > 0.05
    f = run_bench(["--workload", "fof", "--size", "64", "--steps", "1", "--warmup", "0"])
    assert KEYS <= set(f) and f["config"]["groups"] > 0 and f["config"]["particles_in_groups"] > 0
    p = run_bench(["--workload", "hydro", "--sph", "pe", "--size", "32", "--steps", "1", "--warmup", "0"])
    assert KEYS <= set(p) and "pressure-entropy" in p["config"]["workload"]
    d = run_bench(["--workload", "domain", "--size", "64", "--steps", "1", "--warmup", "0"])
    assert KEYS <= set(d) and d["config"]["max_load_over_mean"] < 1.5 and "TopLeaves" in d["config"]["workload"]
