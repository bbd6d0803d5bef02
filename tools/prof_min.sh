#!/bin/bash
# the part of tools/prof.sh that a change of one gravity kernel invalidates: kernel trace of the headline line and the four HBM-traffic
# passes whose sums bench.py reports per library build (profiles/walk_traffic.json, sph_traffic.json).  usage: tools/prof_min.sh <tag>
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras"
HARGS="--workload hydro --steps 2 --warmup 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py $PARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python $ROOT/bench.py $PARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o pmc -- python $ROOT/bench.py $PARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_hydro_fetch -o pmc -- python $ROOT/bench.py $HARGS > /dev/null 2> $OUT/pmc_hydro_fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_hydro_write -o pmc -- python $ROOT/bench.py $HARGS > /dev/null 2> $OUT/pmc_hydro_write.err
cd $ROOT
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name '*.csv' -size +4M -delete
tail -30 $OUT/summary.txt
