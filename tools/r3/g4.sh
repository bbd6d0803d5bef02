set -x
mkdir -p gpurun_out/r3d
timeout 1200 python -m pytest tests/test_gpu_cabi.py tests/test_gpu_gravity.py -x -q -m gpu -k "cabi or c_caller or walk_kernel_variants or list_kernels_agree" > gpurun_out/r3d/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3d/tests.log
tail -25 gpurun_out/r3d/tests.log
for b in 5 6; do
  MPG_LISTS8_BLOCKS=$b timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3d/bench_b$b.json 2> gpurun_out/r3d/bench_b$b.err; echo "blk $b rc=$?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/r3d/bench_b$b.json") if x.startswith("{")]
d=json.loads(l[-1]); print("blk $b", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
done
MPG_LISTS_MODE=2 bash tools/prof.sh r3d_mode2 --no-extras > /dev/null 2>&1
grep -A9 "k_walk_lists8<false" gpurun_out/prof_r3d_mode2/summary.txt | grep -E "avg|VALU|SALU|WAVE_CYCLES|BUSY" | head
