set -x
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_gravity.py -x -q -m gpu -k "walk_kernel_variants or list_kernels_agree or walk_parity or committed or probe" > gpurun_out/r3a/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3a/tests.log
tail -15 gpurun_out/r3a/tests.log
for m in 1 2; do
  MPG_LISTS_MODE=$m timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3a/bench_mode$m.json 2> gpurun_out/r3a/bench_mode$m.err; echo "mode $m rc=$?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/r3a/bench_mode$m.json") if x.startswith("{")]
d=json.loads(l[-1]); print("mode $m", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], {k:(v["ms_per_step"],v["walk_ms"]) for k,v in d.get("other_inputs",{}).items()}, d.get("phases_ms"))
PY
done
cd /tmp && export TMPDIR=/tmp
for m in 1 2; do
MPG_LISTS_MODE=$m rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3a/trace$m -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for m in 1 2; do f=$(find gpurun_out/r3a/trace$m -name "*kernel_stats.csv" | head -1); echo "== mode $m"; head -8 $f | cut -c1-150; done
find gpurun_out/r3a -name '*.csv' -size +2M -delete
