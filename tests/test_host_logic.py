"""Host-side logic that needs no GPU: synthetic IC generators, the row exchange between ranks and the mpg_comm callbacks (gloo, world_size 2 - 4)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sgrid_is_deterministic_and_inside_box(pkg):
    a, m, box = pkg.ics.s_grid(8)
    b, _, _ = pkg.ics.s_grid(8)
    assert np.array_equal(a, b) and a.shape == (512, 3) and m.dtype == np.float32
    assert a.min() >= 0 and a.max() < box
    # first draws of xorshift64 seed 1234567 (SURVEY App. C.5 recipe)
    v = 1234567
    M = (1 << 64) - 1
    us = []
    for _ in range(3):
        v ^= (v << 13) & M
        v ^= v >> 7
        v ^= (v << 17) & M
        us.append((v >> 11) * 2.0 ** -53)
    sp = box / 8
    assert np.allclose(a[0], np.fmod((0.5 + 0.3 * (np.array(us) - 0.5)) * sp + box, box), rtol=0, atol=1e-9)


def test_blocked_xorshift_equals_sequential(pkg):
    n = 70000
    a = pkg.ics.xorshift64_uniform(n)
    v = 1234567
    M = (1 << 64) - 1
    for i in range(n):
        v ^= (v << 13) & M
        v ^= v >> 7
        v ^= (v << 17) & M
        if i % 9973 == 0 or i == n - 1:
            assert a[i] == (v >> 11) * 2.0 ** -53


def test_szel_and_sclust(pkg):
    p, m, box = pkg.ics.s_zel(16)
    assert p.shape == (4096, 3) and p.min() > 0 and p.max() <= box
    q, _, b2 = pkg.ics.s_clust(8)
    assert q.shape == (512, 3) and q.min() >= 0 and q.max() <= b2


def _domain_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import importlib
    pkg = importlib.import_module("mp-gadget_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # personalised exchange with uneven counts: rank r sends (d + 1 + r) rows tagged (r, d) to rank d
    rows, counts = [], []
    for d in range(world):
        c = 0 if d == rank else d + 1 + rank
        counts.append(c)
        rows.append(torch.full((c, 4), float(10 * rank + d), dtype=torch.float64))
    got = pkg.rows.exchange_rows(torch.cat(rows), counts, world)
    exp = torch.cat([torch.full((0 if s == rank else rank + 1 + s, 4), float(10 * s + rank), dtype=torch.float64) for s in range(world)])
    q.put((rank, bool(torch.equal(got, exp))))
    dist.barrier()
    dist.destroy_process_group()


def test_row_exchange_world2_gloo():
    """rows.exchange_rows (the MPI_Alltoallv of exchange.c as domain_peano.PeanoDomain.exchange issues it): uneven counts, gloo"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_domain_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_sgrid_planes_equal_slices_of_the_full_set(pkg):
    n = 24
    full, _, box = pkg.ics.s_grid(n)
    for a, b in ((0, 5), (7, 19), (23, 24)):
        part, m, box2 = pkg.ics.s_grid_planes(n, a, b)
        assert box2 == box and np.array_equal(part, full[a * n * n:b * n * n]) and len(m) == len(part)


def _comm_worker(rank, world, port, q, scale=1):
    """The mpg_comm callbacks of the multi-rank library path (mp-gadget_amd/dist.py::TorchComm) on host buffers over gloo: called
    through the C function pointers exactly as csrc/dist.hip calls them."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    import importlib
    import numpy as np
    pkg = importlib.import_module("mp-gadget_amd")
    D = pkg.dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = D.TorchComm(torch.device("cpu"))
    c = comm.struct
    ok = c.ThisTask == rank and c.NTask == world and c.device_buffers == 0
    i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    # allreduce: float64 SUM, int64 MAX (dtype 0 / 1, op 0 / 1; mpgadget_hip.h)
    a = np.arange(5, dtype=np.float64) + 10.0 * rank
    ok &= c.allreduce(None, a.ctypes.data, 5, 0, 0, 0) == 0
    ok &= bool(np.array_equal(a, world * np.arange(5.0) + 10.0 * sum(range(world))))
    b = np.array([rank, -rank, 7], dtype=np.int64)
    ok &= c.allreduce(None, b.ctypes.data, 3, 1, 1, 0) == 0
    ok &= bool(np.array_equal(b, [world - 1, 0, 7]))
    # alltoall of one int64 per pair
    s = np.array([100 * rank + d for d in range(world)], dtype=np.int64)
    r = np.zeros(world, dtype=np.int64)
    ok &= c.alltoall_i64(None, i64(s), i64(r)) == 0
    ok &= bool(np.array_equal(r, [100 * src + rank for src in range(world)]))
    # alltoallv of bytes: rank r sends scale (r + d + 1) bytes of value 16 r + d to rank d; scale 1: packed blocks, scale > 1:
    # blocks with gaps of 3 bytes between them on both sides (displacements that are not the running sums)
    gap = 0 if scale == 1 else 3
    sb = np.array([scale * (rank + d + 1) for d in range(world)], dtype=np.int64)
    sd = (np.concatenate([[0], np.cumsum(sb)[:-1]]) + gap * np.arange(world)).astype(np.int64)
    rb = np.array([scale * (src + rank + 1) for src in range(world)], dtype=np.int64)
    rd = (np.concatenate([[0], np.cumsum(rb)[:-1]]) + gap * np.arange(world)).astype(np.int64)
    send = np.full(int(sd[-1] + sb[-1]), 255, np.uint8)
    want = np.zeros(int(rd[-1] + rb[-1]), np.uint8)
    for d in range(world):
        send[sd[d]:sd[d] + sb[d]] = 16 * rank + d
    for src in range(world):
        want[rd[src]:rd[src] + rb[src]] = 16 * src + rank
    # piece 2: the exchange cut into pieces of 2 bytes per peer (the path of calls above A2A_MAX_BYTES; RCCL returned garbage beyond
    # 1 GiB per call in round 1).  With scale 3 and 3 - 4 ranks the largest block has 15 - 21 bytes: 8 - 11 pieces, the number being
    # decided by the largest block of ANY rank while most blocks end earlier (empty trailing pieces)
    for piece in (None, 2):
        keep = D.A2A_MAX_BYTES
        if piece:
            D.A2A_MAX_BYTES = piece * world
        recv = np.zeros(len(want), np.uint8)
        ok &= c.alltoallv(None, send.ctypes.data, i64(sb), i64(sd), recv.ctypes.data, i64(rb), i64(rd), 0) == 0
        D.A2A_MAX_BYTES = keep
        ok &= bool(np.array_equal(recv, want))
    # the all-gather pattern of csrc/dist.hip (allgather_host): every peer gets the SAME block (send displacements all 0)
    blk = np.full(8, rank + 1, np.uint8)
    sb2, sd2 = np.full(world, 8, np.int64), np.zeros(world, np.int64)
    rb2, rd2 = np.full(world, 8, np.int64), (8 * np.arange(world)).astype(np.int64)
    recv = np.zeros(8 * world, np.uint8)
    ok &= c.alltoallv(None, blk.ctypes.data, i64(sb2), i64(sd2), recv.ctypes.data, i64(rb2), i64(rd2), 0) == 0
    ok &= bool(np.array_equal(recv, np.repeat(np.arange(1, world + 1, dtype=np.uint8), 8)))
    ok &= not comm.errors
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_mpg_comm_callbacks_world2_gloo():
    """N > 1 on CPU: the three collectives the library asks its caller for (allreduce, alltoall_i64, alltoallv of bytes) as
    TorchComm implements them, two gloo ranks, host buffers, called through the mpg_comm function pointers."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


@pytest.mark.parametrize("world", [3, 4])
def test_mpg_comm_piecewise_alltoallv_gloo(world):
    """TorchComm's piece-splitting of a large exchange with an artificially small limit: 3 and 4 gloo ranks, unequal blocks with gaps,
    8 - 11 pieces per exchange (the first real multi-GPU RCCL run takes this path for every exchange above 512 MiB)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q, 3)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_bench_gpus_n_refuses_without_n_gpus():
    """`python bench.py --gpus 2` as a plain command on a box that shows fewer than 2 GPUs (here: none) exits with code 2 and prints no
    JSON line - it never falls through to a one-GPU measurement.  A launcher whose WORLD_SIZE disagrees with
    --gpus is refused as well."""
    import json
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs the command asks for")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MPG_DIST_BACKEND", "MPG_FORCE_MGPU")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "32", "--steps", "1", "--warmup", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 2, (r.returncode, r.stderr[-1500:])
    assert "needs 2 visible GPUs" in r.stderr
    assert not [x for x in r.stdout.splitlines() if x.startswith("{")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE 4 != --gpus 2" in r.stderr
