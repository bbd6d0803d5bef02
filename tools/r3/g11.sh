mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests/test_gpu_gravity.py -x -q -m gpu -k "walk_kernel_variants or list_kernels_agree or walk_parity or probe or committed or full_size_256" > gpurun_out/r3k/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3k/tests.log
tail -4 gpurun_out/r3k/tests.log
for b in 6 4 8; do
MPG_STREAM_BLOCKS=$b timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3k/bench_$b.json 2> gpurun_out/r3k/bench_$b.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3k/bench_$b.json") if x.startswith("{")][-1])
print("blocks $b ms/step", d["ms_per_step"], "walk", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"])
PY
done
bash tools/prof.sh r3k --no-extras > /dev/null 2>&1
grep -E "k_walk_lists8<false|k_walk_stream" gpurun_out/prof_r3k/summary.txt | head -4; grep "steady" gpurun_out/prof_r3k/summary.txt
grep -A12 "k_walk_stream<true, true, true, 6> *dispatches" gpurun_out/prof_r3k/summary.txt | head -40
