#!/bin/bash
# chunks of 8 targets per wave of the two walk kernels (MPG_SPLIT_CPW; 0 = persistent grids): same-box A/B on the headline set
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$1 step %.2f ms walk %.2f ms' % (j['ms_per_step'], r['avg_launch_ms']))"; }
for c in ${CPWS:-2 1 2 1}; do MPG_SPLIT_CPW=$c run cpw_$c; done
