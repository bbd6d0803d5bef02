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
    assert "workload" in j["config"] and "s_zel" in j["config"]["workload"]          # the headline set of SURVEY 8(d)
    r = j["roofline"]
    assert r["bound"] == "fp64_valu" and r["unit"] == "TFLOP/s" and 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # roofline.traffic is measured in the run itself (two child processes under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): the walk's lists
    # alone are 4 B per entry written once and read once
    assert r["traffic"] and "measured in this run" in r["traffic_note"], r["traffic_note"]
    assert r["traffic"] > 8 * (r["leaf_entries_per_launch"] + r["node_entries_per_launch"]) and 0 < r["hbm_measured_frac"] < 1
    # the walk's two kernels one by one (an event between them inside the timed region)
    k = r["kernels_ms"]
    assert k["launches_timed"] == j["steps"] and k["k_walk_lists8"] > 0 and k["k_walk_eval"] > 0
    assert abs(k["k_walk_lists8"] + k["k_walk_eval"] - r["avg_launch_ms"]) < 0.05 * r["avg_launch_ms"] + 0.05
    cb = j["cpu_baseline"]
    assert cb["physical_cores"] >= 1 and len(cb["walk_s_all"]) == 3 and cb["cpu_model"] and cb["processes"] >= 1
    assert cb["cores"] == sum(cb["threads_per_process"]) and cb["pairs_per_s_per_thread"] > 1e6 and cb["tree_build_own_share_s"] >= 0
    assert set(j["other_inputs"]) == {"s_grid", "s_clust"} and all(v["ms_per_step"] > 0 for v in j["other_inputs"].values())
    assert j["host_path"]["ms_per_step"] > 0 and set(j["host_path"]["calls_ms"]) == {"gravpm_force", "force_tree_full", "grav_short_tree"}
    # the device-resident drop-in mode: the same calls without per-call transfers, and the same physics
    rp = j["resident_path"]
    assert 0 < rp["ms_per_step"] < j["host_path"]["ms_per_step"] and rp["mean_abs_accel"] > 0
    # SURVEY 8(d) metric (ii): the short-range-only sub-steps, on the tree of all particles and on the tree of the active ones
    assert set(j["substeps"]) == {"1/8", "1/64", "1/512"}
    for v in j["substeps"].values():
        assert v["all_particle_tree"]["ms_per_substep"] > 0 and v["active_only_tree"]["ms_per_substep"] > 0 and v["active"] >= 1
    # BASELINE configs[2] in the default line, graded against the fp64 vector peak
    h = j["hydro"]
    assert h["ms_per_step"] > 0 and h["roofline"]["bound"] == "fp64_valu" and 0 < h["roofline"]["frac"] <= 1 and 0 < h["roofline_hydro"]["frac"] <= 1


@pytest.mark.parametrize("overlap", [True, False])
def test_multi_gpu_path_in_a_one_rank_group_checks_itself(overlap):
    """bench.py --gpus N runs the library's choreography and then checks its own forces: sampled particles recomputed on one GPU from
    the whole set (parity_check in the line; a failed check exits non-zero).  mpg_dist_gravity_step builds the local tree on a second
    thread and stream beside the PM step; MPG_DIST_NO_OVERLAP=1 runs the phases one after the other: both forms pass the same check."""
    env = {"MPG_FORCE_MGPU": "1", "MASTER_PORT": "29171" if overlap else "29172"}
    if not overlap:
        env["MPG_DIST_NO_OVERLAP"] = "1"
    j = run_bench(["--gpus", "1", "--size", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=env)
    assert KEYS <= set(j) and j["value"] > 1e6
    pc = j["parity_check"]
    assert pc["ok"] and pc["counters_equal"] and pc["n"] >= 1024 and pc["median_rel"] <= 1e-12 and pc["gravpm_max_rel_to_mean"] <= 1e-11
    assert (j["phases_ms"]["dist_tree_build_beside_pm_ms"] > 0) == overlap
    # the collectives ran on the library's native RCCL communicator (csrc/rccl_comm.hip), not through Python callbacks
    assert j["config"]["communicator"].startswith("native RCCL"), j["config"]["communicator"]
    cc = j["config"]["communicator_calls"]
    assert cc["alltoallv"] > 0 and cc["allreduce"] > 0 and cc["alltoall_i64"] > 0 and cc["rccl_version"] > 20000


@pytest.mark.parametrize("mode", ["self_through_rccl", "torch_callbacks"])
def test_multi_gpu_path_communicator_variants(mode):
    """self_through_rccl: MPG_RCCL_SELF=1 sends the rank's own block through ncclSend / ncclRecv as well (on real peers only the other
    ranks' blocks travel that way, the own block is a device copy), with pieces of 64 KiB so that every block is cut: on a one-GPU box
    this is what exercises the grouped send / receive path with data.  torch_callbacks: the torch.distributed callbacks of rounds 2-3
    (--comm torch) still pass the same self-check."""
    env = {"MPG_FORCE_MGPU": "1", "MASTER_PORT": "29173"}
    args = ["--gpus", "1", "--size", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    if mode == "self_through_rccl":
        env.update(MPG_RCCL_SELF="1", MPG_RCCL_PIECE="65536")
    else:
        args += ["--comm", "torch"]
    j = run_bench(args, env=env)
    pc = j["parity_check"]
    assert pc["ok"] and pc["counters_equal"] and pc["median_rel"] <= 1e-12, pc
    assert j["config"]["communicator"].startswith("native RCCL" if mode == "self_through_rccl" else "torch.distributed")


def test_multi_gpu_parity_check_on_four_ranks():
    """... and on 4 ranks (gloo, sharing this GPU), clustered set, after the rebalancing exchange: the self-check the driver's 8-GPU
    run carries (BASELINE configs[4]: "per-step force tolerance check")."""
    env = dict(os.environ, MPG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port",
           "29874", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--size", "48", "--ic", "s_clust", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    pc = j["parity_check"]
    assert pc["ok"] and pc["counters_equal"] and pc["n"] >= 2000, pc


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` as a PLAIN command (no torchrun, no WORLD_SIZE - the shape of the driver's N = 1 command): bench.py
    launches the two ranks itself.  With the default backend it must REFUSE on this one-GPU box (exit code 2, no JSON line: an N-GPU
    command never prints a line measured on fewer GPUs); with MPG_DIST_BACKEND=gloo (ranks sharing the GPU) it prints `n_gpus: 2`
    with a green parity_check and one roofline entry per rank."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MPG_DIST_BACKEND", "MPG_FORCE_MGPU")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 2, (r.returncode, r.stderr[-2000:])
        assert "needs 2 visible GPUs" in r.stderr and not [x for x in r.stdout.splitlines() if x.startswith("{")], (r.stdout[-500:], r.stderr[-500:])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, MPG_DIST_BACKEND="gloo"), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert KEYS <= set(j) and j["n_gpus"] == 2 and j["config"]["particles"] == 64 ** 3
    assert j["parity_check"]["ok"] and j["parity_check"]["counters_equal"], j["parity_check"]
    pr = j["roofline"]["per_rank"]
    assert len(pr["walk_ms"]) == 2 and len(pr["frac"]) == 2 and sum(pr["own_particles"]) == 64 ** 3 and all(0 < f <= 1 for f in pr["frac"])
    assert j["config"]["communicator"].startswith("torch.distributed") and j["config"]["rccl_ranks"] is None
    # a launcher whose rank count disagrees with --gpus is an error, not a line
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE 1 != --gpus 2" in r.stderr


def test_one_rank_rccl_group_reports_the_communicator_rccl_sees():
    j = run_bench(["--gpus", "1", "--size", "32", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env={"MPG_FORCE_MGPU": "1", "MASTER_PORT": "29174"})
    assert j["config"]["rccl_ranks"] == 1 and j["n_gpus"] == 1 and len(j["roofline"]["per_rank"]["walk_ms"]) == 1


def test_other_workloads():
    h = run_bench(["--workload", "hydro", "--size", "32", "--steps", "1", "--warmup", "0"])
    assert KEYS <= set(h) and "roofline_hydro" in h and h["roofline"]["bound"] == "fp64_valu"
    sb = run_bench(["--workload", "substep", "--size", "64", "--steps", "2", "--active-frac", "0.125"])
    assert KEYS <= set(sb) and sb["config"]["active"] == 64 ** 3 // 8 and sb["value"] > 0
    i = run_bench(["--workload", "integrate", "--size", "64", "--steps", "2", "--warmup", "1"])
    assert KEYS <= set(i) and i["roofline"]["frac"] > 0.05
    f = run_bench(["--workload", "fof", "--size", "64", "--steps", "1", "--warmup", "0"])
    assert KEYS <= set(f) and f["config"]["groups"] > 0 and f["config"]["particles_in_groups"] > 0
    p = run_bench(["--workload", "hydro", "--sph", "pe", "--size", "32", "--steps", "1", "--warmup", "0"])
    assert KEYS <= set(p) and "pressure-entropy" in p["config"]["workload"]
    d = run_bench(["--workload", "domain", "--size", "64", "--steps", "1", "--warmup", "0"])
    assert KEYS <= set(d) and d["config"]["max_load_over_mean"] < 1.5 and "TopLeaves" in d["config"]["workload"]


@pytest.mark.parametrize("ic", ["s_zel", "s_clust"])
def test_c4_shape_through_rccl_at_full_per_gpu_size(ic):
    """BASELINE configs[3] (512^3 on 8 GPUs) as ONE rank sees it: 256^3 own particles, the distributed choreography of mpg_dist_*
    with every collective issued through RCCL (a one-rank group: MPG_FORCE_MGPU) on device buffers - the decomposition, the particle
    shipping of the PM, the transposes, the ghost import and the all-reduce of the top of the tree at their full per-GPU sizes."""
    j = run_bench(["--gpus", "1", "--size", "256", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--ic", ic],
                  env={"MPG_FORCE_MGPU": "1", "MASTER_PORT": "29871"})
    assert KEYS <= set(j) and j["value"] > 2e7 and j["config"]["particles"] == 256 ** 3
    assert j["roofline"]["pp_interactions_per_launch"] > 0 and j["phases_ms"]["dist_transpose_bytes"] > 2 * 512 ** 3 * 8


def test_c4_whole_particle_set_on_one_gpu():
    """BASELINE configs[3]'s WHOLE particle set (512^3 particles, Nmesh 1024) on the one GPU: 288 GB hold the set, its 1024^3 mesh and
    the walk lists; the walk runs its 64-bit-offset kernels (source and node arrays beyond 4 GiB) and slices its list area.  One timed
    step; the line's counters must be those of a 512^3 Zel'dovich walk (about 700 pair interactions and 200 nodes used per target)
    and the rate that of the 256^3 run.  (this run was the builder's, not part of -m gpu.)"""
    j = run_bench(["--size", "512", "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline"])   # (the warm-up step sizes the walk's lists)
    assert KEYS <= set(j) and j["config"]["particles"] == 512 ** 3 and j["config"]["nmesh"] == 1024
    r = j["roofline"]
    assert j["value"] > 1.2e8 and 0.08 < r["frac"] <= 1
    assert 400 < r["pp_interactions_per_launch"] / 512 ** 3 < 1200 and 100 < r["nodes_used_per_launch"] / 512 ** 3 < 400
    assert r["targets_per_launch"] == 512 ** 3 and r["walk_variant"] == 6


def test_c5_whole_particle_set_on_one_gpu():
    """BASELINE configs[4]'s WHOLE particle set (2 x 256^3 DM + gas, pressure-entropy SPH) on the one GPU: gravity, gas tree, density,
    hmax, hydro force of 16.8 M gas particles; one Hsml pass per step once converged, about 110 neighbours per gas particle (quintic
    kernel, eta = 1)."""
    j = run_bench(["--workload", "hydro", "--size", "256", "--sph", "pe", "--steps", "1", "--warmup", "1"])
    assert j["config"]["particles"] == 2 * 256 ** 3 and "pressure-entropy" in j["config"]["workload"] and j["value"] > 4e7
    assert j["config"]["density_iterations"][-1] == 1
    import re
    ngb = int(re.search(r"\((\d+) neighbours", j["roofline"]["note"]).group(1)) / 256 ** 3
    assert 90 < ngb < 130, ngb


def test_c5_shape_through_rccl_at_full_per_gpu_size():
    """BASELINE configs[4] (2 x 256^3 pressure-entropy hydro on 8 GPUs) as ONE rank sees it: 2 x 128^3 own particles, gravity + the
    distributed SPH loops (ghost import, the ghosts' SPH fields refreshed from their owners between density and hydro) with the
    collectives on RCCL (one-rank group)."""
    j = run_bench(["--workload", "hydro", "--gpus", "1", "--size", "128", "--steps", "1", "--warmup", "1", "--sph", "pe"],
                  env={"MPG_FORCE_MGPU": "1", "MASTER_PORT": "29872"})
    assert KEYS - {"roofline"} <= set(j) and j["config"]["particles"] == 2 * 128 ** 3 and "pressure-entropy" in j["config"]["workload"]
    assert j["value"] > 5e6
    pc = j["parity_check"]        # the line checks its own SPH results against the one-GPU loops (configs[4]: "per-step force tolerance check")
    assert pc["ok"] and pc["n"] >= 1024 and pc["hsml_frac_within_1e-12"] >= 0.999 and pc["hydroaccel_frac_within_1e-10"] >= 0.999, pc


@pytest.mark.parametrize("sph", ["pe", "de"])
def test_multi_gpu_hydro_parity_check_on_four_ranks(sph):
    """bench.py --workload hydro --gpus 4 (gloo ranks sharing this GPU): after the timed steps the ranks' density() + hydro_force() are
    repeated on ONE GPU over the whole set from the same starting smoothing lengths and sampled gas targets compared (Hsml 1e-12,
    Density / HydroAccel / DtEntropy 1e-10, loop counters); a failed check exits with code 3.  Both SPH formulations."""
    env = dict(os.environ, MPG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port",
           "29875" if sph == "pe" else "29876", os.path.join(ROOT, "bench.py"), "--workload", "hydro", "--gpus", "4", "--size", "32", "--sph", sph,
           "--steps", "1", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads(r.stdout.strip().splitlines()[-1])
    pc = j["parity_check"]
    assert pc["ok"] and pc["n"] >= 1500 and pc["hsml_max_rel"] <= 1e-3, pc
    assert pc["counters_equal"] or all(abs(x - y) <= 1e-5 * y for x, y in zip(pc["counters_ranks"], pc["counters_one_gpu"])), pc


def test_peano_domains_balance_the_walk_work_on_the_clustered_set():
    """4 ranks (gloo, sharing this GPU) on the strongly clustered set: the TopLeaves dealt out by the measured work per particle
    (domain.c:611) even out the walk's work; equal particle numbers do not."""
    env = dict(os.environ, MPG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port",
           "29873", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--size", "64", "--ic", "s_clust", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--overdecomp", "32", "--no-parity-check"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    lb = j["config"]["load_balance"]
    assert lb["walk_work_max_over_mean"] < 1.15, lb
    assert lb["by_particle_number"]["walk_work_max_over_mean"] > 1.2 * lb["walk_work_max_over_mean"], lb
