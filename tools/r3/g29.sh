mkdir -p gpurun_out/r3ac
for ov in 0 1; do
E=""; [ $ov = 0 ] && E="MPG_DIST_NO_OVERLAP=1"
env $E MPG_FORCE_MGPU=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2957$ov RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3ac/bench_$ov.json 2> gpurun_out/r3ac/bench_$ov.err
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3ac/bench_$ov.json") if x.startswith("{")][-1])
print("overlap $ov ms/step", d["ms_per_step"], {k:v for k,v in d["phases_ms"].items() if k.startswith("dist")}, d.get("parity_check"))
PY
done
timeout 1200 python -m pytest tests/test_gpu_bench.py tests/test_gpu_cabi.py -x -q -m gpu > gpurun_out/r3ac/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3ac/tests.log
tail -4 gpurun_out/r3ac/tests.log
