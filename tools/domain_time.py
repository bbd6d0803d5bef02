"""Time of the device passes of the Peano-Hilbert domain decomposition on one GPU at n^3 particles, for NTASK tasks: the sample
(keys of all particles + strided gather + sort) and the TopLeaf / Task / count pass; the host tree arithmetic in between.
usage: python tools/domain_time.py [n=256] [ntask=8]"""
import ctypes as C, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mp-gadget_amd")
DP = importlib.import_module("mp-gadget_amd.domain_peano")
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ntask = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pos, mass, box = pkg.ics.s_clust(n, seed=5) if os.environ.get("MPG_IC") == "s_clust" else pkg.ics.s_zel(n)
N = len(pos)
d_pos = torch.from_numpy(pos).cuda()
eng = pkg.Engine(0)
eng.use_torch_stream()
lib = eng.lib
P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
pol = DP.DomainPolicy(0, ntask)
keys = np.zeros(N // pol.SubSampleDistance + 2, np.uint64)
ns = C.c_int64(0)
res = {}
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert lib.mpg_dev_domain_sample(eng.h, C.c_int64(N), C.c_void_p(d_pos.data_ptr()), None, C.c_double(box), pol.PreSort, pol.SubSampleDistance,
                                     P(keys, C.c_uint64), C.c_int64(len(keys)), C.byref(ns)) == 0
    t1 = time.perf_counter()
    maxn = int(0.5 * (N + 1))
    tree = np.zeros(min(maxn, 1 << 20) + 8, DP.TOPNODE_DTYPE)
    size, failed = C.c_int(0), C.c_int(0)
    assert lib.mpg_domain_local_refine(P(keys, C.c_uint64), None, ns, P(tree, DP.TopNode), C.byref(size), len(tree) - 8, C.byref(failed)) == 0 and not failed.value
    lim = int(tree[0]["Count"]) // pol.NTopLeaves
    lib.mpg_domain_toptree_truncate(P(tree, DP.TopNode), C.byref(size), C.c_int64(lim), C.c_int64(lim))
    lib.mpg_domain_global_refine(P(tree, DP.TopNode), C.byref(size), len(tree) - 8, C.c_int64(lim), C.c_int64(lim), C.byref(failed))
    ltn = np.zeros(size.value, np.int32); nl = C.c_int(0)
    lib.mpg_domain_create_topleaves(P(tree, DP.TopNode), size, P(ltn, C.c_int), C.byref(nl))
    t2 = time.perf_counter()
    counts = np.zeros(nl.value, np.int64)
    assert lib.mpg_dev_domain_topleaves(eng.h, C.c_int64(N), C.c_void_p(d_pos.data_ptr()), None, C.c_double(box), P(tree, DP.TopNode), size, nl,
                                        None, ntask, None, None, P(counts, C.c_int64), None) == 0
    t3 = time.perf_counter()
    lt = np.zeros(nl.value, np.int32); st = np.zeros(ntask, np.int32); en = np.zeros(ntask, np.int32)
    assert lib.mpg_domain_assign_topleaves_balanced(P(tree, DP.TopNode), size, P(ltn, C.c_int), nl, P(counts, C.c_int64), ntask, 1, P(lt, C.c_int), P(st, C.c_int),
                                                    P(en, C.c_int)) == 0
    t4 = time.perf_counter()
    tl = torch.zeros(N, dtype=torch.int32, device="cuda"); ta = torch.zeros(N, dtype=torch.int32, device="cuda")
    fc = np.zeros(nl.value, np.int64); tc = np.zeros(ntask, np.int64)
    torch.cuda.synchronize(); t5 = time.perf_counter()
    assert lib.mpg_dev_domain_topleaves(eng.h, C.c_int64(N), C.c_void_p(d_pos.data_ptr()), None, C.c_double(box), P(tree, DP.TopNode), size, nl,
                                        P(lt, C.c_int), ntask, C.c_void_p(tl.data_ptr()), C.c_void_p(ta.data_ptr()), P(fc, C.c_int64), P(tc, C.c_int64)) == 0
    t6 = time.perf_counter()
    res = dict(particles=N, ntask=ntask, nsample=ns.value, topnodes=size.value, topleaves=nl.value, sample_ms=(t1 - t0) * 1e3, host_tree_ms=(t2 - t1) * 1e3,
               count_pass_ms=(t3 - t2) * 1e3, assign_ms=(t4 - t3) * 1e3, layout_pass_ms=(t6 - t5) * 1e3,
               max_load_over_mean=float(tc.max() / tc.mean()), GBps_layout_pass=N * 32 / (t6 - t5) / 1e9)
print(json.dumps(res))
eng.close()
