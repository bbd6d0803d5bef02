#!/bin/bash
# Stress of the point at which the one multi-rank failure of round 4 happened ("send counts and live rows unequal on one
# rank" in the decomposition + exchange of a 2-rank helper's set-up): tools/flake_stress2.sh <iterations> [particles]
# Every iteration starts FRESH processes - 2, 3 and 4 gloo ranks sharing the GPU in turn - that decompose and exchange one particle set through
# BOTH choreographies (Python PeanoDomain and the library's mpg_dist_domain_*), with MPG_POISON=1 (fresh device allocations filled with 0xFF)
# and the invariants of round 6 armed (PeanoDomain.exchange's assertion, mpg_dist_domain_exchange's count check).  Nothing is retried.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
N=${1:-200}
NP=${2:-60000}
mkdir -p gpurun_out/flake
export MPG_POISON=1 MPG_DIST_BACKEND=gloo MPG_GLOBAL_SORT=1 PYTHONPATH=$ROOT
fail=0
t0=$(date +%s)
for i in $(seq 1 $N); do
  ranks=$((2 + i % 3))
  port=$((29600 + i % 200))
  out=/tmp/flake2_$i
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks --master-addr 127.0.0.1 --master-port $port \
      tools/mgpu_domain_check.py $out $NP > gpurun_out/flake/stress2_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "ITERATION $i ($ranks ranks) FAILED rc=$rc"; tail -40 gpurun_out/flake/stress2_$i.log; fail=$((fail+1)); break; fi
  rm -f gpurun_out/flake/stress2_$i.log $out.*.npz
done
t1=$(date +%s)
echo "stress2 done: $i iterations of $N, $fail failed, $((t1-t0)) s, $NP particles, MPG_POISON=1, ranks 2/3/4 in turn"
