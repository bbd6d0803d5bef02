#!/bin/bash
# same-box A/B of grav_walk_split.hip build variants on two input sets (GPU box): tools/walk_ab.sh <out> "<flags1>" "<flags2>" ...
# ("" = the default build).  Prints per variant and input set: ms per step, walk ms, frac, the per-kernel split (MPG_SPLIT_TIME) and the
# counters; the default build is restored at the end.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$1; shift
mkdir -p $(dirname $OUT)
: > $OUT
for f in "$@"; do
    if [ -n "$f" ]; then export MPG_EXTRA_FLAGS="grav_walk_split.hip:$f"; else unset MPG_EXTRA_FLAGS; fi
    python mp-gadget_amd/build.py > /dev/null 2>&1 || echo "build failed: $f" | tee -a $OUT
    for ic in ${ICS:-s_zel s_clust}; do
        python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras --ic $ic 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('[%s] %s step %.2f ms  walk %.2f ms  frac %.4f  pp %d nodes_used %d leaf_entries %s node_entries %s fallback %s' % ('$f', '$ic', j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['pp_interactions_per_launch'], r['nodes_used_per_launch'], r.get('leaf_entries_per_launch'), r.get('node_entries_per_launch'), r.get('targets_to_fallback_kernel')))" | tee -a $OUT
        MPG_SPLIT_TIME=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --ic $ic 2>&1 >/dev/null | grep SPLIT_TIME | tail -3 | sed "s/^/[$f] $ic /" | tee -a $OUT
    done
done
unset MPG_EXTRA_FLAGS
python mp-gadget_amd/build.py > /dev/null 2>&1
