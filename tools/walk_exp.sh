#!/bin/bash
# builds grav_walk_split.hip with experiment flags ($1) and prints the walk time of the default bench: tools/walk_exp.sh "-DX=1" (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
if [ -n "$1" ]; then export MPG_EXTRA_FLAGS="grav_walk_split.hip:$1"; else unset MPG_EXTRA_FLAGS; fi
python mp-gadget_amd/build.py > /dev/null 2>&1
python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags [$1]: step', round(j['ms_per_step'],2), 'walk', round(j['roofline']['avg_launch_ms'],2))"
