#!/usr/bin/env python3
"""Register / LDS / occupancy figures of every kernel of one csrc file, from hipcc's -Rpass-analysis=kernel-resource-usage
(cross-compiles for gfx950: runs without a GPU).  usage: tools/kernel_resources.py grav_walk_split.hip [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "mp-gadget_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: +([A-Za-z ]+(?:\[[A-Za-z/]*\])?): *(.*?)(?: \[-Rpass|$)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k in ("Function Name", "Name"):
        if cur and flt in cur.get("name", ""):
            print(cur)
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(mpg::TreeView.*", "", name).replace("void mpg::(anonymous namespace)::", "")[-70:]}
    else:
        cur[k] = v
if cur and flt in cur.get("name", ""):
    print(cur)
