#!/bin/bash
# pair kernel: bit identity and a profile (kernel trace + SQ counters) to see where its time goes
mkdir -p gpurun_out/pair
cd /root/repo
for ic in s_zel s_clust; do
  for pr in 0 1; do
    MPG_LISTS_PAIR=$pr python tools/pair_check.py gpurun_out/pair/${ic}_$pr.npz $ic 64 > gpurun_out/pair/${ic}_$pr.log 2>&1 || tail -5 gpurun_out/pair/${ic}_$pr.log
  done
  python - <<PY
import numpy as np
a=np.load("gpurun_out/pair/${ic}_0.npz"); b=np.load("gpurun_out/pair/${ic}_1.npz")
for k in a.files:
    same = np.array_equal(a[k], b[k])
    d = np.abs(a[k]-b[k]).max()
    print("$ic", k, "bit-identical" if same else "DIFFERENT max|d| %g (of %g), %d entries" % (d, np.abs(a[k]).max(), (a[k]!=b[k]).sum()))
PY
done
ROOT=/root/repo; OUT=$ROOT/gpurun_out/prof_pair; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MPG_LISTS_PAIR=1 MPG_LISTS_BLOCKS=5
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq -o pmc -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o pmc -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq2.err
cd $ROOT
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name '*.csv' -size +4M -delete
grep -A14 "kernel stats" $OUT/summary.txt | head -12; grep -B1 -A9 "k_walk_lists2<false" $OUT/summary.txt | head -60
