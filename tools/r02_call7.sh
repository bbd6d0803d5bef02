#!/bin/bash
# round 2, GPU call 7: FOF across ranks (fixed), walk kernel occupancy / packing with the serial single-slice walk
mkdir -p gpurun_out/c7
export MASTER_ADDR=127.0.0.1
timeout 900 python -m pytest tests/test_gpu_fof.py::test_fof_groups_spanning_ranks "tests/test_gpu_gravity.py::test_walk_kernel_variants" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/c7/new.log 2>&1; echo "new rc=$? $(tail -1 gpurun_out/c7/new.log)"
grep -E "^FAILED|^ERROR" gpurun_out/c7/new.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline $IC > gpurun_out/c7/$name.json 2>/dev/null; }
IC=""
run zel_default X=1
run zel_pack MPG_PACK_LEAVES=1
run zel_l8 MPG_LISTS_BLOCKS=8
run zel_e5 MPG_EVAL_BLOCKS=5
run zel_e4 MPG_EVAL_BLOCKS=4
run zel_l8_e5 MPG_LISTS_BLOCKS=8 MPG_EVAL_BLOCKS=5
run zel_l8_e4_pack MPG_LISTS_BLOCKS=8 MPG_EVAL_BLOCKS=4 MPG_PACK_LEAVES=1
IC="--ic s_grid"
run grid_default X=1
run grid_l8_e5 MPG_LISTS_BLOCKS=8 MPG_EVAL_BLOCKS=5
run grid_pack MPG_PACK_LEAVES=1
IC="--ic s_clust"
run clust_default X=1
run clust_ov1 MPG_SPLIT_OVERLAP=1
run clust_l8_e5 MPG_LISTS_BLOCKS=8 MPG_EVAL_BLOCKS=5
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c7/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), round(d['roofline']['frac'],4))
    except Exception as e: print(f, 'ERR', e)
PY
