#!/bin/bash
# same-box, same-build A/B of the fp32 pre-classification in k_walk_lists8 (GPU box): MPG_LISTS_F32=0 keeps the fp64 tests.
# usage: tools/lists_f32_ab.sh <out> ; prints per input set and setting: ms per step, walk ms, frac, the two kernels' times, counters
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$1
mkdir -p $(dirname $OUT)
: > $OUT
for ic in ${ICS:-s_zel s_clust s_grid}; do
    for f32 in 0 1; do
        MPG_LISTS_F32=$f32 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras --ic $ic 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; k=r.get('kernels_ms',{}); print('[f32=$f32] $ic step %.2f ms  walk %.2f ms  frac %.4f  lists %s eval %s  pp %d nodes_used %d visited %d fp64_passes %s node_steps %s fallback %s' % (j['ms_per_step'], r['avg_launch_ms'], r['frac'], k.get('k_walk_lists8'), k.get('k_walk_eval'), r['pp_interactions_per_launch'], r['nodes_used_per_launch'], r['nodes_visited_per_launch'], r.get('fp32_fallback_passes_per_launch'), r.get('node_steps_per_launch'), r.get('targets_to_fallback_kernel')))" | tee -a $OUT
    done
done
