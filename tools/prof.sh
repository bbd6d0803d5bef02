#!/bin/bash
# rocprofv3 recipe for the bench (run on the GPU box through gpurun; outputs under gpurun_out/prof_*)
# usage: tools/prof.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic $*"   # (bench.py would start rocprofv3 children of its own: not under a profiler)
# (the counter passes run the headline workload alone: the traffic per walk is summed over all walks of the run, and the sub-step /
# other-input legs of the default line are walks of other sizes)
PARGS="$ARGS --no-extras"
# 1. kernel trace + stats
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
# 2. PMC passes (separate runs, no tracing domains)
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq -o pmc -- python $ROOT/bench.py $PARGS > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o pmc -- python $ROOT/bench.py $PARGS > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python $ROOT/bench.py $PARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o pmc -- python $ROOT/bench.py $PARGS > /dev/null 2> $OUT/pmc_write.err
# 3. the SPH kernels of configs[2] (bench.py --workload hydro): trace + HBM traffic passes
HARGS="--workload hydro --steps 2 --warmup 1 --no-live-traffic"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_hydro -o trace -- python $ROOT/bench.py $HARGS > $OUT/bench_hydro.json 2> $OUT/trace_hydro.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_hydro_fetch -o pmc -- python $ROOT/bench.py $HARGS > /dev/null 2> $OUT/pmc_hydro_fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_hydro_write -o pmc -- python $ROOT/bench.py $HARGS > /dev/null 2> $OUT/pmc_hydro_write.err
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_hydro_sq -o pmc -- python $ROOT/bench.py $HARGS > /dev/null 2> $OUT/pmc_hydro_sq.err
cd $ROOT
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name '*.csv' -size +4M -delete   # (per-dispatch traces / counter dumps of tens of MB; the stats CSV and the summary are what is kept)
cat $OUT/summary.txt
