#!/bin/bash
# builds sph.hip with experiment flags ($1) and prints the kernel times of the hydro bench: tools/sph_exp.sh "-DSPH_EXP_X" (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
if [ -n "$1" ]; then export MPG_EXTRA_FLAGS="sph.hip:$1"; fi
python mp-gadget_amd/build.py > /dev/null 2>&1
echo "== flags: $1"
timeout 300 bash tools/prof_any.sh python bench.py --workload hydro --steps 3 --warmup 1 2>&1 | grep -i "k_density\|k_hydro\|k_sph" 
