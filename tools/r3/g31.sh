mkdir -p gpurun_out/r3ae
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--workload hydro --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/r3ae/pmc1 -o pmc -- python $R/bench.py $A > /dev/null 2> $R/gpurun_out/r3ae/pmc1.err
rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $R/gpurun_out/r3ae/pmc2 -o pmc -- python $R/bench.py $A > /dev/null 2> $R/gpurun_out/r3ae/pmc2.err
rocprofv3 --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $R/gpurun_out/r3ae/pmc3 -o pmc -- python $R/bench.py $A > /dev/null 2> $R/gpurun_out/r3ae/pmc3.err
python - <<PY
import csv, glob, collections
for d in ("pmc1","pmc2","pmc3"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$R/gpurun_out/r3ae/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            kn=r["Kernel_Name"]
            if "k_density" in kn or "k_hydro(" in kn:
                acc[kn[:14]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k, {n: "max %.4g n %d" % (max(x), len(x)) for n,x in v.items()})
PY
find $R/gpurun_out/r3ae -name "*counter_collection.csv" -size +2M -delete
