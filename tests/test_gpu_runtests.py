"""The reference's own end-to-end acceptance test of the force path, run_gravity_test (libgadget/runtests.c:89-232), on this engine
through the snapshot wire format: tools/run_gravity_test.py reads an IC written in the reference's format, runs pairs / open tree /
tree / Rcut 9.5 / Nmesh/2 and applies the reference's thresholds (:147, :180, :197, :217); its PART-* outputs are read back."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_gravity_test_thresholds(pkg, tmp_path):
    snap = importlib.import_module("mp-gadget_amd.snapshot")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    rgt = importlib.import_module("run_gravity_test")
    n = 32
    pos, mass, box = pkg.ics.s_zel(n)
    N = len(pos)
    ic = str(tmp_path / "IC")
    snap.write_snapshot(ic, {1: dict(Position=pos, Velocity=np.zeros((N, 3)), ID=np.arange(N, dtype=np.uint64))}, box, 0.1,
                        mass_table=[0, float(mass[0]), 0, 0, 0, 0])
    out = str(tmp_path / "out")
    rep = rgt.main([ic, out, "--nmesh", str(2 * n)])
    assert rep["open_vs_pairs"][1] <= 0.1 and rep["tree_vs_open"][0] <= 1.2 * 0.002 and rep["rcut"][0] <= 0.002
    assert rep["nmesh2"][0] >= rep["tree_vs_open"][0] and rep["nmesh2"][1] >= rep["tree_vs_open"][1]
    # the fully open tree is the pairwise force up to the cube / sphere difference at the cut-off: far better than the bound
    assert 0 < rep["open_vs_pairs"][0] < 1e-3 and rep["rcut"] != rep["tree_vs_open"]
    for name in ("PART-pairs", "PART-tree-open", "PART-tree", "PART-tree-rcut", "PART-tree-nmesh2"):
        p = os.path.join(out, name + "-000")
        info = snap.block_info(p, "1/GravAccel")
        assert info == dict(dtype="<f4", nmemb=3, nfile=1, size=N) and snap.block_info(p, "1/GravPM")["size"] == N
    a = snap.read_block(os.path.join(out, "PART-tree-000"), "1/GravAccel", dtype="f8")
    b = snap.read_block(os.path.join(out, "PART-tree-open-000"), "1/GravAccel", dtype="f8")
    assert np.isfinite(a).all() and 0 < np.abs(a - b).max() < 0.05 * np.abs(b).max()
