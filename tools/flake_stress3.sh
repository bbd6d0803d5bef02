#!/bin/bash
# Stress of the helper in which the round-4 failure happened (tools/mgpu_hydro_check.py, 2 gloo ranks on one GPU, its set-up: Peano-Hilbert
# decomposition + exchange, then density -> hydro_force through mpg_dist_*): tools/flake_stress3.sh <iterations>
# Fresh processes every iteration (2, 3, 4 ranks in turn), MPG_POISON=1, nothing retried.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
N=${1:-100}
mkdir -p gpurun_out/flake
export MPG_POISON=1 MPG_DIST_BACKEND=gloo MPG_MGPU_MODE=peano MASTER_ADDR=127.0.0.1 PYTHONPATH=$ROOT
fail=0
t0=$(date +%s)
for i in $(seq 1 $N); do
  ranks=$((2 + i % 3))
  port=$((29800 + i % 150))
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks --master-addr 127.0.0.1 --master-port $port \
      tools/mgpu_hydro_check.py /tmp/flake3_$i.npz 24 > gpurun_out/flake/stress3_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "ITERATION $i ($ranks ranks) FAILED rc=$rc"; tail -40 gpurun_out/flake/stress3_$i.log; fail=$((fail+1)); break; fi
  rm -f gpurun_out/flake/stress3_$i.log /tmp/flake3_$i.npz
done
t1=$(date +%s)
echo "stress3 done: $i iterations of $N, $fail failed, $((t1-t0)) s, mgpu_hydro_check.py 24^3 x 2 species, MPG_POISON=1, ranks 2/3/4 in turn"
