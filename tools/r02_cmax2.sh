#!/bin/bash
cd /root/repo
for ic in s_grid s_clust s_zel; do
 for cfg in "1 5" "0 6"; do
  set -- $cfg
  echo "== $ic MPG_LISTS_PAIR=$1 MPG_LISTS_BLOCKS=$2"
  MPG_LISTS_PAIR=$1 MPG_LISTS_BLOCKS=$2 python bench.py --ic $ic --no-extras --no-cpu-baseline --steps 8 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
 done
done
