#!/bin/bash
# repeat the multi-rank tests that put 3-4 gloo ranks on one GPU until one fails (the unclosed flake of DESIGN section 4): tools/flake_stress.sh <iterations>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
N=${1:-20}
mkdir -p gpurun_out/flake
for i in $(seq 1 $N); do
  timeout 600 python -m pytest tests/test_gpu_sph.py tests/test_gpu_domain.py tests/test_gpu_gravity.py tests/test_gpu_fof.py -m gpu -x -q -p no:cacheprovider \
      -k "sph_peano_ranks_match_one or decomposition_and_exchange_on_ranks or peano_domain_ranks_match_one or groups_spanning_ranks" > gpurun_out/flake/stress_$i.log 2>&1
  rc=$?
  tail -1 gpurun_out/flake/stress_$i.log
  if [ $rc -ne 0 ]; then echo "ITERATION $i FAILED rc=$rc"; tail -60 gpurun_out/flake/stress_$i.log; break; fi
  rm -f gpurun_out/flake/stress_$i.log
done
echo "stress done: $i iterations"
