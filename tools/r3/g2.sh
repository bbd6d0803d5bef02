# PMC comparison of the list kernels (modes 1 and 2), headline set
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for m in 1 2; do
  MPG_LISTS_MODE=$m bash tools/prof.sh r3a_mode$m --no-extras > /dev/null 2>&1
  cp gpurun_out/prof_r3a_mode$m/summary.txt gpurun_out/r3a/summary_mode$m.txt
  python - <<PY
import json
l=[x for x in open("gpurun_out/prof_r3a_mode$m/bench_trace.json") if x.startswith("{")]
r=json.loads(l[-1])["roofline"]; print("mode $m steps", r["node_steps_per_launch"], "lanes", r["node_lanes_per_launch"], "visited", r["nodes_visited_per_launch"])
PY
done
cat gpurun_out/r3a/summary_mode2.txt
