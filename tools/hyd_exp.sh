#!/bin/bash
# gravity part of the hydro line with experiment flags for grav_walk_split.hip: tools/hyd_exp.sh "-DX" (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
if [ -n "$1" ]; then export MPG_EXTRA_FLAGS="grav_walk_split.hip:$1"; else unset MPG_EXTRA_FLAGS; fi
python mp-gadget_amd/build.py > /dev/null 2>&1
python bench.py --workload hydro --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags [$1]: step', round(j['ms_per_step'],2), j['phases_ms'])"
