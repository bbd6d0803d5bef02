"""A C caller of the C-ABI (tests/c/test_cabi.c): compiled with plain gcc against include/mpgadget_hip.h and linked with
libmpgadget_hip.so, the way the reference's run.c would be.  The build is checked on CPU; the runs need the GPU.

 * single: the drop-in (host pointer) calls on 160-byte struct particle_data records against the committed vectors of tests/golden/;
 * ranks:  2 / 4 forked processes whose mpg_comm callbacks are plain C on shared memory (stand-ins for MPI_Allreduce / MPI_Alltoall /
           MPI_Alltoallv), each with its own engine, through mpg_dist_set_domain / mpg_dist_gravity_step - against the same vectors
           and, on a larger Zel'dovich set, against the CPU oracle."""
import importlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = 43.0071


def build(tmp_path):
    exe = str(tmp_path / "test_cabi")
    lib = os.path.join(ROOT, "mp-gadget_amd")
    cmd = ["gcc", "-O2", "-std=gnu11", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "test_cabi.c"), "-o", exe,
           "-L", lib, "-lmpgadget_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-lm", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_c_program_builds_against_the_header(tmp_path):
    """gcc compiles the C caller against include/mpgadget_hip.h (a C header: no C++ in the signatures) and links every symbol it
    uses from the library."""
    importlib.import_module("__graft_entry__").build()
    assert os.path.exists(build(tmp_path))


def _write_case(tmp_path, name, pos, expect_gpm, expect_acc):
    pos.astype("<f8").tofile(str(tmp_path / (name + ".pos")))
    np.concatenate([expect_gpm.ravel(), expect_acc.ravel()]).astype("<f8").tofile(str(tmp_path / (name + ".expect")))
    return str(tmp_path / (name + ".pos")), str(tmp_path / (name + ".expect"))


def _run(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PASS" in r.stdout.splitlines()[-1], (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.gpu
def test_c_caller_single_and_ranks_against_golden(tmp_path):
    pkg = importlib.import_module("mp-gadget_amd")
    exe = build(tmp_path)
    table = os.path.join(ROOT, "mp-gadget_amd", "data", "shortrange_force_kernels.f64")
    g = np.load(os.path.join(ROOT, "tests", "golden", "grav_sgrid16.npz"))
    pos, mass, box = pkg.ics.s_grid(16)
    assert box == float(g["box"])
    p, e = _write_case(tmp_path, "sgrid16", pos, g["GravPM"], g["Accel2"])
    _run(exe, "single", table, p, e, 16, 32, box)
    for nt in (1, 2, 4):
        out = _run(exe, "ranks", table, p, e, 16, 32, box, nt)
        assert "PASS ranks %d" % nt in out
    _run(exe, "ranks_host", table, p, e, 16, 32, box, 2)


@pytest.mark.gpu
def test_c_caller_ranks_against_oracle(tmp_path, orc):
    """a set large enough for a real decomposition level (Rcut = 9 of 64 mesh cells: La = 2), Zel'dovich-displaced: 2 and 4 C ranks"""
    from oracle import oracle as O
    pkg = importlib.import_module("mp-gadget_amd")
    exe = build(tmp_path)
    table = os.path.join(ROOT, "mp-gadget_amd", "data", "shortrange_force_kernels.f64")
    n, nmesh = 32, 64
    pos, mass, box = pkg.ics.s_zel(n)
    gpm, _ = O.gravpm_force(pos, mass, box, nmesh, 1.5, G)
    tr = orc.tree(pos, mass, box)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 1
    a1, _, _, _ = tr.grav_short_tree(par, oldacc=np.sqrt((gpm ** 2).sum(1)) / G)
    par.TreeUseBH = 0
    a2, _, _, _ = tr.grav_short_tree(par, oldacc=np.sqrt(((a1 + gpm) ** 2).sum(1)) / G)
    p, e = _write_case(tmp_path, "szel32", pos, gpm, a2)
    for nt in (2, 4):
        _run(exe, "ranks", table, p, e, n, nmesh, box, nt)
    _run(exe, "ranks_host", table, p, e, n, nmesh, box, 4)
