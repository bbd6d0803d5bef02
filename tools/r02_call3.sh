#!/bin/bash
# round 2, GPU call 3: the new distributed path (mpg_dist_*), the C caller, full-size config tests; walk priority experiment
mkdir -p gpurun_out/c3
export MASTER_ADDR=127.0.0.1
timeout 1500 python -m pytest tests/test_gpu_cabi.py "tests/test_gpu_gravity.py::test_peano_domain_ranks_match_one" "tests/test_gpu_gravity.py::test_rccl_one_rank_group_matches_single" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/c3/dist.log 2>&1; echo "dist rc=$? $(tail -1 gpurun_out/c3/dist.log)"
grep -E "^FAILED|^ERROR" gpurun_out/c3/dist.log
timeout 1500 python -m pytest tests/test_gpu_bench.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/c3/bench_tests.log 2>&1; echo "bench tests rc=$? $(tail -1 gpurun_out/c3/bench_tests.log)"
grep -E "^FAILED|^ERROR" gpurun_out/c3/bench_tests.log
timeout 1500 python -m pytest "tests/test_gpu_gravity.py::test_full_size_256_properties" "tests/test_gpu_sph.py::test_full_size_hydro_2x128" tests/test_gpu_sph.py::test_reference_density_known_answer_on_gpu tests/test_gpu_gravity.py::test_reference_force_accuracy_vs_direct_sum_on_gpu -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/c3/fullsize.log 2>&1; echo "fullsize rc=$? $(tail -1 gpurun_out/c3/fullsize.log)"
grep -E "^FAILED|^ERROR" gpurun_out/c3/fullsize.log
for pack in 0 1; do for lp in 0 2 3; do
  MPG_PACK_LEAVES=$pack MPG_LIST_PRIO=$lp timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c3/zel_p${pack}_l${lp}.json 2>/dev/null
done; done
MPG_PACK_LEAVES=1 MPG_EVAL_PRIO=2 timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c3/zel_p1_e2.json 2>/dev/null
MPG_PACK_LEAVES=0 MPG_EVAL_PRIO=2 timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c3/zel_p0_e2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c3/zel_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), round(d['roofline']['frac'],4))
    except Exception as e: print(f, 'ERR', e)
PY
