#!/bin/bash
# round 2, GPU call 4: fixed tests, distributed SPH, slice-size experiment, three full suites
mkdir -p gpurun_out/c4
export MASTER_ADDR=127.0.0.1
timeout 1200 python -m pytest "tests/test_gpu_gravity.py::test_peano_domain_ranks_match_one" tests/test_gpu_sph.py::test_sph_peano_ranks_match_one "tests/test_gpu_sph.py::test_full_size_hydro_2x128" tests/test_gpu_bench.py::test_default_workload_line tests/test_gpu_bench.py::test_peano_domains_balance_the_walk_work_on_the_clustered_set -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/c4/fixed.log 2>&1; echo "fixed rc=$? $(tail -1 gpurun_out/c4/fixed.log)"
grep -E "^FAILED|^ERROR" gpurun_out/c4/fixed.log
for sl in 262144 524288 1048576 2097152 4194304; do
  MPG_SPLIT_SLICE=$sl timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c4/zel_slice_$sl.json 2>/dev/null
done
MPG_SPLIT_SLICE=524288 timeout 300 python bench.py --ic s_grid --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c4/grid_slice_524288.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c4/*_slice_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), round(d['roofline']['frac'],4))
    except Exception as e: print(f, 'ERR', e)
PY
tools/flake_hunt.sh 3
