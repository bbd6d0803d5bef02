#!/bin/bash
cd /root/repo
for cfg in "0 0" "0 1" "1 0"; do
  set -- $cfg
  echo "== MPG_LISTS_PAIR=$1 MPG_PACK_LEAVES=$2"
  MPG_LISTS_PAIR=$1 MPG_PACK_LEAVES=$2 MPG_LISTS_BLOCKS=$([ $1 = 1 ] && echo 5 || echo 6) python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
cd /tmp; export TMPDIR=/tmp
MPG_PACK_LEAVES=1 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/tr -o trace -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
grep -i "k_walk" /tmp/tr/*/trace_kernel_stats.csv /tmp/tr/trace_kernel_stats.csv 2>/dev/null | cut -c1-160
