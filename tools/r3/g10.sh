mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests/test_gpu_gravity.py -x -q -m gpu -k "walk_kernel_variants or list_kernels_agree or walk_parity or probe or committed or full_size_256" > gpurun_out/r3j/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3j/tests.log
tail -4 gpurun_out/r3j/tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3j/bench.json 2> gpurun_out/r3j/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3j/bench.json") if x.startswith("{")][-1])
print("ms/step", d["ms_per_step"], "walk", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"])
print({k:(v["ms_per_step"],v["walk_ms"]) for k,v in d["other_inputs"].items()}, d["hydro"]["ms_per_step"])
PY
MPG_LISTS_MODE=2 bash tools/prof.sh r3j --no-extras > /dev/null 2>&1
grep -E "k_walk_lists8<false|k_walk_eval" gpurun_out/prof_r3j/summary.txt | head -4; grep "steady" gpurun_out/prof_r3j/summary.txt
grep -A9 "k_walk_eval<true, true, true, true, 6> *dispatches" gpurun_out/prof_r3j/summary.txt | grep -E "VALU|BUSY" | head
