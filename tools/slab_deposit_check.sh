#!/bin/bash
# one-rank slab PM (forced collectives) on the clustered set: deposit modes
for m in plain sorted auto; do
  if [ $m = auto ]; then unset MPG_PM_DEPOSIT; else export MPG_PM_DEPOSIT=$m; fi
  MPG_FORCE_MGPU=1 timeout 400 python bench.py --ic s_clust --n 128 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['ms_per_step'],1), d['phases_ms'].get('pm_slab_total_incl_collectives'))"
done
