mkdir -p gpurun_out/r3i
timeout 1800 python -m pytest tests/test_gpu_gravity.py tests/test_gpu_cabi.py tests/test_gpu_sph.py tests/test_gpu_fof.py tests/test_gpu_timestep.py tests/test_gpu_domain.py tests/test_gpu_bench.py -x -q -m gpu -k "ranks or peano or rccl or c_caller or multi_gpu or c4 or c5 or spanning or evol or default_workload or other_workloads" --durations=10 > gpurun_out/r3i/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3i/tests.log
tail -22 gpurun_out/r3i/tests.log
MPG_FORCE_MGPU=1 MASTER_PORT=29871 timeout 600 python bench.py --gpus 1 --size 256 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3i/c4.json 2> gpurun_out/r3i/c4.err; echo rc=$?
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3i/c4.json") if x.startswith("{")][-1])
print("ms/step", d["ms_per_step"], "walk", d["roofline"]["avg_launch_ms"]); print({k:v for k,v in d["phases_ms"].items() if k.startswith("dist")}); print(d["parity_check"]["ok"])
PY
