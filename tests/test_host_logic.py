"""Host-side logic that needs no GPU: synthetic IC generators and the target-sharding exchange (gloo, world_size 2)."""
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


def test_slot_ranges_partition(pkg):
    for n in (0, 1, 7, 1000, 4097):
        for w in (1, 2, 3, 8):
            r = [pkg.shard.slot_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(h - l for l, h in r) <= pkg.shard.chunk_size(n, w)


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    import importlib
    pkg = importlib.import_module("mp-gadget_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    order = torch.randperm(n, generator=g).to(torch.int32)          # same on every rank
    truth = torch.randn(n, 3, dtype=torch.float64, generator=g)     # what a single rank would compute
    lo, hi = pkg.shard.slot_range(n, rank, world)
    vals = torch.full((n, 3), float("nan"), dtype=torch.float64)
    mine = order[lo:hi].long()
    vals[mine] = truth[mine]                                        # this rank "walked" only its own targets
    pkg.shard.exchange_results(vals, order, rank, world)
    q.put((rank, bool(torch.equal(vals, truth))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1001, 64])
def test_exchange_results_world2_gloo(n):
    """N>1 path on CPU: two gloo ranks each own half of the tree-order slots; after the all-gather both hold the
    single-rank answer bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _slab_worker(rank, world, port, q):
    """Collectives and index logic of the slab-decomposed PM on CPU tensors (gloo): the all-to-all transpose of a toy
    [x][y] array, the ring pass of ghost planes, slab ownership of particles and the per-target all-gather."""
    sys.path.insert(0, ROOT)
    import importlib
    pkg = importlib.import_module("mp-gadget_amd")
    S = pkg.pm_slab
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    # transpose: rank r holds rows x in [r P, (r+1) P) of A[x][y]; after the all-to-all it holds columns y in its range, x slowest
    M, P = 8, 8 // world
    A = torch.arange(M * M, dtype=torch.float64).reshape(M, M)
    mine = A[rank * P:(rank + 1) * P]                                    # [P][M]
    send = torch.stack([mine[:, d * P:(d + 1) * P] for d in range(world)]).contiguous()   # [d][xl][yl]
    recv = torch.empty_like(send)
    S._all_to_all(recv.view(-1), send.view(-1), world)
    ok &= bool(torch.equal(recv.reshape(M, P), A[:, rank * P:(rank + 1) * P]))
    # the same through the piecewise path (calls above A2A_MAX_ELEMENTS elements are split: RCCL fails above 1 GiB)
    keep, S.A2A_MAX_ELEMENTS = S.A2A_MAX_ELEMENTS, 5 * world
    recv2 = torch.full_like(send, -1.0)
    S._all_to_all(recv2.view(-1), send.view(-1), world)
    S.A2A_MAX_ELEMENTS = keep
    ok &= bool(torch.equal(recv2, recv))
    # ghost planes of the potential: first 3 planes to the previous rank, last 2 to the next (the same rank when world == 2)
    class _E:   # the exchange only needs the buffers
        pass
    spm = S.SlabPM.__new__(S.SlabPM)
    spm.world, spm.rank, spm.group = world, rank, None
    spm.ghost_send = torch.stack([torch.full((6,), 10.0 * rank + k, dtype=torch.float64) for k in range(5)])
    spm.ghost_recv = torch.zeros_like(spm.ghost_send)
    spm._ghost_planes()
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    exp = [10.0 * nxt + 0, 10.0 * nxt + 1, 10.0 * nxt + 2, 10.0 * prv + 3, 10.0 * prv + 4]
    ok &= bool(torch.equal(spm.ghost_recv[:, 0], torch.tensor(exp, dtype=torch.float64)))
    # slab ownership incl. x == box (wraps to cell 0) and the per-target exchange
    nmesh, box = 16, 4.0
    gen = torch.Generator().manual_seed(3)
    n = 1000
    pos = torch.rand(n, 3, dtype=torch.float64, generator=gen) * box
    pos[0, 0] = box
    owner = S.slab_of_cells(pos[:, 0], box / nmesh, nmesh, world)
    ok &= int(owner[0]) == 0 and int(owner.min()) >= 0 and int(owner.max()) < world
    order = torch.randperm(n, generator=gen).to(torch.int32)
    tg = order[owner[order.long()] == rank].contiguous()
    truth = torch.randn(n, 3, dtype=torch.float64, generator=gen)
    vals = torch.full((n, 3), float("nan"), dtype=torch.float64)
    vals[tg.long()] = truth[tg.long()]
    S.TargetExchange(world, torch.device("cpu")).exchange(vals, tg)
    ok &= bool(torch.equal(vals, truth))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_slab_pm_collectives_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_slab_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_tree_columns_and_needed_sets(pkg):
    """Index logic of the distributed-particle mode: tree_column replays the octree descent (root cell 1.001 Box around Box/2);
    every rank's needed columns cover its slab widened by the margin, periodically."""
    D = pkg.domain
    box, La = 8.0, 3
    x = torch.tensor([1e-9, 0.9959, 0.9961, 3.999, 4.001, 7.99, 8.0], dtype=torch.float64)
    w, lo = 1.001 * box / 8, box / 2 - 0.5 * 1.001 * box
    ref = torch.floor((x - lo) / w).to(torch.int64)
    assert torch.equal(D.tree_column(x, box, La), ref)
    need = D.needed_columns(box, 64, 4, La, margin=0.7)
    assert need.shape == (4, 8)
    for s in range(4):
        for xx in np.linspace(s * 2.0 - 0.7, (s + 1) * 2.0 + 0.7, 50):
            xx = float(np.mod(xx, box))
            assert bool(need[s, int(np.floor((xx - lo) / w))]), (s, xx)
    assert bool(need[0, 7]) and bool(need[3, 0])            # periodic neighbours across the box edge
    assert not bool(need[0, 3]) and not bool(need[2, 0])


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
    got = pkg.pm_slab.exchange_rows(torch.cat(rows), counts, world)
    exp = torch.cat([torch.full((0 if s == rank else rank + 1 + s, 4), float(10 * s + rank), dtype=torch.float64) for s in range(world)])
    q.put((rank, bool(torch.equal(got, exp))))
    dist.barrier()
    dist.destroy_process_group()


def test_ghost_exchange_world2_gloo():
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


def _migrate_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import importlib
    pkg = importlib.import_module("mp-gadget_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    box, nmesh = 8.0, 32
    dom = pkg.domain.SlabDomain(None, box, nmesh, rank, world, torch.device("cpu"), rcut=1.0)
    g = torch.Generator().manual_seed(11)
    pos = torch.rand(3000, 3, dtype=torch.float64, generator=g) * box      # the same global set on every rank
    ids = torch.arange(3000, dtype=torch.float64)
    # 64-bit IDs as the reference's stars / black holes carry them: Generation in bits 56+ (not representable in a double),
    # one byte-wide and one float32 column as well: every column must arrive bit for bit
    big = (torch.arange(3000, dtype=torch.int64) * 2654435761 + 12345) | (torch.arange(3000, dtype=torch.int64) % 5 + 1 << 56) | 1
    typ8 = (torch.arange(3000) % 6).to(torch.uint8)
    m32 = torch.rand(3000, dtype=torch.float32, generator=g)
    own = dom.select_own(pos)
    p, i = pos[own].clone(), ids[own].clone()
    p[:, 0] = torch.remainder(p[:, 0] + 1.7, box)                          # a "drift" that carries many particles across slab faces
    p2, i2, big2, typ2, m2 = dom.migrate(p, (i, big[own].clone(), typ8[own].clone(), m32[own].clone()))
    assert big2.dtype == torch.int64 and typ2.dtype == torch.uint8 and m2.dtype == torch.float32
    okb = bool(torch.equal(big2, big[i2.long()]) and torch.equal(typ2, typ8[i2.long()]) and torch.equal(m2, m32[i2.long()]))
    owner = pkg.pm_slab.slab_of_cells(p2[:, 0], box / nmesh, nmesh, world)
    ok = bool((owner == rank).all())
    # global check: every id exactly once, carried with its own position
    cnt = torch.zeros(3000, dtype=torch.float64)
    cnt[i2.long()] += 1
    dist.all_reduce(cnt)
    ok &= bool((cnt == 1).all())
    exp = pos[i2.long()].clone()
    exp[:, 0] = torch.remainder(exp[:, 0] + 1.7, box)
    ok &= bool(torch.equal(exp, p2)) and okb
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_particle_migration_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_migrate_worker, args=(r, 2, port, q)) for r in range(2)]
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


def _comm_worker(rank, world, port, q):
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
    # alltoallv of bytes: rank r sends (r + d + 1) bytes of value 16 r + d to rank d, packed
    sb = np.array([rank + d + 1 for d in range(world)], dtype=np.int64)
    sd = np.concatenate([[0], np.cumsum(sb)[:-1]]).astype(np.int64)
    rb = np.array([src + rank + 1 for src in range(world)], dtype=np.int64)
    rd = np.concatenate([[0], np.cumsum(rb)[:-1]]).astype(np.int64)
    send = np.concatenate([np.full(sb[d], 16 * rank + d, np.uint8) for d in range(world)])
    want = np.concatenate([np.full(rb[src], 16 * src + rank, np.uint8) for src in range(world)])
    for piece in (None, 2):      # 2: the exchange cut into pieces of 2 bytes per peer (the path of calls above A2A_MAX_BYTES)
        keep = D.A2A_MAX_BYTES
        if piece:
            D.A2A_MAX_BYTES = piece * world
        recv = np.zeros(int(rb.sum()), np.uint8)
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
