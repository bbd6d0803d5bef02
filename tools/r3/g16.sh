mkdir -p gpurun_out/r3p
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "0 1024" "2 1280"; do
set -- $cfg
MPG_LEAF_EXPAND=$1 MPG_LIST_CAP=$2 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/r3p/pmc1_$1 -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/r3p/pmc1_$1.err
MPG_LEAF_EXPAND=$1 MPG_LIST_CAP=$2 rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $R/gpurun_out/r3p/pmc2_$1 -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/r3p/pmc2_$1.err
MPG_LEAF_EXPAND=$1 MPG_LIST_CAP=$2 rocprofv3 --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum -d $R/gpurun_out/r3p/pmc3_$1 -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/r3p/pmc3_$1.err
python - <<PY
import csv, glob, collections
for d in ("pmc1_$1","pmc2_$1","pmc3_$1"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$R/gpurun_out/r3p/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            kn=r["Kernel_Name"]
            if "k_walk_eval<" in kn or "k_walk_lists8<false" in kn:
                acc[kn[28:52]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print("kx $1", k, {n: "%.4g"%(sum(x[1:])/max(len(x)-1,1)) for n,x in v.items()})
PY
done
find $R/gpurun_out/r3p -name "*counter_collection.csv" -size +2M -delete
tail -3 $R/gpurun_out/r3p/pmc3_0.err
