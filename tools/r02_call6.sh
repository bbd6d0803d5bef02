#!/bin/bash
# round 2, GPU call 6: FOF across ranks, hydro test fix, slice / overlap experiment, then full suites
mkdir -p gpurun_out/c6
export MASTER_ADDR=127.0.0.1
timeout 900 python -m pytest tests/test_gpu_fof.py "tests/test_gpu_sph.py::test_full_size_hydro_2x128" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/c6/new.log 2>&1; echo "new rc=$? $(tail -1 gpurun_out/c6/new.log)"
grep -E "^FAILED|^ERROR" gpurun_out/c6/new.log
for v in "8388608 1" "16777216 1" "8388608 0" "4194304 0"; do set -- $v
  MPG_SPLIT_SLICE=$1 MPG_SPLIT_OVERLAP=$2 timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c6/zel_slice_$1_ov$2.json 2>/dev/null
done
MPG_SPLIT_SLICE=16777216 timeout 300 python bench.py --ic s_grid --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c6/grid_slice_16777216_ov1.json 2>/dev/null
timeout 300 python bench.py --ic s_grid --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c6/grid_default.json 2>/dev/null
timeout 300 python bench.py --ic s_clust --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c6/clust_default.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c6/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), round(d['roofline']['frac'],4))
    except Exception as e: print(f, 'ERR', e)
PY
tools/flake_hunt.sh 2
