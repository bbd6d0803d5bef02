#!/bin/bash
# round 2, GPU call 5: library decomposition / evolution on Peano domains, C caller with the real decomposition, hunt for the rare
# abort of a rank (complete stderr kept), hydro diagnostics, slice size, rocprofv3 profile of the headline bench
mkdir -p gpurun_out/c5
export MASTER_ADDR=127.0.0.1
timeout 900 python -m pytest tests/test_gpu_cabi.py tests/test_gpu_domain.py tests/test_gpu_timestep.py::test_distributed_evolution_matches_one_gpu "tests/test_gpu_sph.py::test_full_size_hydro_2x128[0]" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/c5/new.log 2>&1; echo "new rc=$? $(tail -1 gpurun_out/c5/new.log)"
grep -E "^FAILED|^ERROR" gpurun_out/c5/new.log
for i in 1 2 3 4 5; do
  timeout 900 python -m pytest tests/test_gpu_sph.py::test_sph_peano_ranks_match_one tests/test_gpu_sph.py::test_sph_ranks_match_one "tests/test_gpu_gravity.py::test_peano_domain_ranks_match_one" tests/test_gpu_gravity.py::test_two_ranks_match_one tests/test_gpu_domain.py::test_decomposition_and_exchange_on_ranks -m gpu -q --tb=line -p no:cacheprovider > gpurun_out/c5/ranks_$i.log 2>&1; echo "ranks $i rc=$? $(tail -1 gpurun_out/c5/ranks_$i.log)"
done
for sl in 8388608; do
  MPG_SPLIT_SLICE=$sl timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c5/zel_slice_$sl.json 2>/dev/null
done
timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c5/zel_default.json 2>/dev/null
timeout 300 python bench.py --ic s_grid --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c5/grid_default.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c5/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), round(d['roofline']['frac'],4))
    except Exception as e: print(f, 'ERR', e)
PY
tools/prof.sh r02c_szel --no-extras > gpurun_out/c5/prof.log 2>&1; tail -5 gpurun_out/c5/prof.log
