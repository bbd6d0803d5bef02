#!/bin/bash
# round 2, GPU call 2: gloo ordering check, new tests + packing, bench line, 3 more full suites, then the suite with gloo's own all_to_all
mkdir -p gpurun_out/c2
export MASTER_ADDR=127.0.0.1
for np in 2 4; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port 29611 tools/gloo_a2a_check.py 150 > gpurun_out/c2/gloo_check_$np.log 2>&1
  echo "gloo check np=$np rc=$?"; grep "wrong results" gpurun_out/c2/gloo_check_$np.log
done
timeout 900 python -m pytest tests/test_gpu_gravity.py tests/test_gpu_sph.py -m gpu -x -q -p no:cacheprovider > gpurun_out/c2/quick.log 2>&1; echo "quick rc=$? $(tail -1 gpurun_out/c2/quick.log)"
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/c2/bench_szel.json 2> gpurun_out/c2/bench_szel.err; echo "bench rc=$?"; tail -c 600 gpurun_out/c2/bench_szel.json
MPG_PACK_LEAVES=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/c2/bench_szel_nopack.json 2>/dev/null; echo "nopack rc=$?"
for ic in s_grid; do
  timeout 300 python bench.py --ic $ic --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/c2/bench_$ic.json 2>/dev/null
  MPG_PACK_LEAVES=0 timeout 300 python bench.py --ic $ic --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/c2/bench_${ic}_nopack.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c2/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), round(d['roofline']['frac'],4))
    except Exception as e: print(f, 'ERR', e)
PY
tools/flake_hunt.sh 3
mkdir -p gpurun_out/flake_old && mv gpurun_out/flake/run_*.log gpurun_out/flake_old/ 2>/dev/null
MPG_GLOO_TRY_A2A=1 tools/flake_hunt.sh 4
