#!/bin/bash
# The round-4 multi-rank failure happened once, on a box's FIRST multi-process use.  This script is meant to be the first GPU work of a fresh
# box (one gpurun call per run): the 2-rank helper of that failure (tools/mgpu_hydro_check.py, Peano-Hilbert decomposition + exchange, then
# density -> hydro_force through mpg_dist_*) before anything has touched the device, then 3 and 4 ranks; no poison fill, nothing retried.
# tools/flake_cold.sh <tag>  ->  one line in gpurun_out/flake_cold/<tag>.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
TAG=${1:-0}
mkdir -p gpurun_out/flake_cold
export MPG_DIST_BACKEND=gloo MPG_MGPU_MODE=peano MASTER_ADDR=127.0.0.1 PYTHONPATH=$ROOT
res=""
for ranks in 2 3 4; do
  t0=$(date +%s.%N)
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks --master-addr 127.0.0.1 --master-port $((29700 + ranks)) \
      tools/mgpu_hydro_check.py /tmp/cold_$ranks.npz 24 > gpurun_out/flake_cold/${TAG}_r$ranks.log 2>&1
  rc=$?
  t1=$(date +%s.%N)
  res="$res ranks=$ranks rc=$rc $(python -c "print('%.1f s' % ($t1 - $t0))");"
  if [ $rc -eq 0 ]; then rm -f gpurun_out/flake_cold/${TAG}_r$ranks.log; else tail -30 gpurun_out/flake_cold/${TAG}_r$ranks.log; fi
done
echo "cold box $TAG ($(hostname)):$res" | tee gpurun_out/flake_cold/$TAG.txt
