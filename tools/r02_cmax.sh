#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_gravity.py -m gpu -q -x -k "variants or walk_parity or committed or accuracy or pair_list or full_size_256" 2>&1 | tail -4
for cfg in "0 6" "1 5" "1 4"; do
  set -- $cfg
  echo "== MPG_LISTS_PAIR=$1 MPG_LISTS_BLOCKS=$2"
  MPG_LISTS_PAIR=$1 MPG_LISTS_BLOCKS=$2 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
for ic in s_grid s_clust; do
  echo "== $ic"
  python bench.py --ic $ic --no-extras --no-cpu-baseline --steps 6 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
