#!/bin/bash
# CIC deposit: plain atomics vs cell-sorted wave-aggregated, per input set (MPG_PM_DEPOSIT), and what the auto-selection picks
for ic in s_grid s_zel s_clust; do
  for m in plain sorted auto; do
    if [ $m = auto ]; then unset MPG_PM_DEPOSIT; else export MPG_PM_DEPOSIT=$m; fi
    timeout 300 python bench.py --ic $ic --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ic', '$m', 'deposit', d['phases_ms']['pm_deposit'], 'pm', d['phases_ms']['pm_total'], 'step', round(d['ms_per_step'],1))"
  done
done
