set -x
mkdir -p gpurun_out/r3c
timeout 900 python -m pytest tests/test_gpu_gravity.py -x -q -m gpu -k "walk_kernel_variants or list_kernels_agree or walk_parity" > gpurun_out/r3c/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3c/tests.log
tail -5 gpurun_out/r3c/tests.log
for b in 4 5 6; do
  MPG_LISTS8_BLOCKS=$b timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3c/bench_b$b.json 2> gpurun_out/r3c/bench_b$b.err; echo "blk $b rc=$?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/r3c/bench_b$b.json") if x.startswith("{")]
d=json.loads(l[-1]); print("blk $b", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
done
MPG_LISTS_MODE=2 bash tools/prof.sh r3c_mode2 --no-extras > /dev/null 2>&1
head -40 gpurun_out/prof_r3c_mode2/summary.txt
grep -A9 "k_walk_lists8<false" gpurun_out/prof_r3c_mode2/summary.txt | head -60
