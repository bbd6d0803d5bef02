#!/bin/bash
# pair-target list kernel (MPG_LISTS_PAIR): bit-identity with the one-target kernel, the walk tests, and timings
mkdir -p gpurun_out/pair
cd /root/repo
for ic in s_zel s_clust s_grid; do
  for pr in 0 1; do
    MPG_LISTS_PAIR=$pr python tools/pair_check.py gpurun_out/pair/${ic}_$pr.npz $ic 64 > gpurun_out/pair/${ic}_$pr.log 2>&1 || tail -5 gpurun_out/pair/${ic}_$pr.log
  done
  python - <<PY
import numpy as np
a=np.load("gpurun_out/pair/${ic}_0.npz"); b=np.load("gpurun_out/pair/${ic}_1.npz")
for k in a.files:
    same = np.array_equal(a[k], b[k])
    d = np.abs(a[k]-b[k]).max()
    print("$ic", k, "bit-identical" if same else "DIFFERENT max|d| %g (of %g), %d entries" % (d, np.abs(a[k]).max(), (a[k]!=b[k]).sum()))
PY
done
timeout 1500 python -m pytest tests/test_gpu_gravity.py -m gpu -q -x -k "variants or parity or committed or accuracy or domain_ranks" 2>&1 | tail -5
for cfg in "0 6" "1 4" "1 5" "1 6"; do
  set -- $cfg
  echo "== MPG_LISTS_PAIR=$1 MPG_LISTS_BLOCKS=$2"
  MPG_LISTS_PAIR=$1 MPG_LISTS_BLOCKS=$2 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
for ic in s_grid s_clust; do
 for pr in 0 1; do
  echo "== $ic MPG_LISTS_PAIR=$pr"
  MPG_LISTS_PAIR=$pr python bench.py --ic $ic --no-extras --no-cpu-baseline --steps 6 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
 done
done
