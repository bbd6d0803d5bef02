#!/usr/bin/env python3
"""Extract the calibrated TreePM short-range window table as a DATA file.

The reference carries its "exact" short-range force window as a 512x5 table of
numbers (libgadget/shortrange-kernel.c, consumed by gravity.c:22-51).  The table
was calibrated numerically against a brute-force PM computation and cannot be
regenerated from a closed form, so the numbers (not the source text) are
carried as a little-endian float64 binary: 512 rows x 5 columns
  [x (mesh cells), w_pot, w_force, erfc_pot, erfc_force].

Run in the build container only (it reads /root/reference); the output file
mp-gadget_amd/data/shortrange_force_kernels.f64 is committed and travels.
"""
import re, sys, os
import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/libgadget/shortrange-kernel.c"
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(
    os.path.dirname(__file__), "..", "mp-gadget_amd", "data", "shortrange_force_kernels.f64")
rows = []
for line in open(src):
    line = line.strip()
    if not line.startswith("{"):
        continue
    nums = re.findall(r"[-+]?\d+\.\d+e[-+]\d+", line)
    if len(nums) == 5:
        rows.append([float(x) for x in nums])
tab = np.asarray(rows, dtype="<f8")
assert tab.shape == (512, 5), tab.shape
assert tab[0, 0] == 0.0 and abs(tab[-1, 0] - 15.0) < 1e-12
tab.tofile(out)
print("wrote", out, tab.shape, "dx =", repr(tab[1, 0]))
