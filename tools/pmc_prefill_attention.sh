# Counter passes over the running-softmax prompt attention kernel: bash tools/pmc_prefill_attention.sh TAG SEQS LEN
# (each pass its own run; the bench tool also runs the partitioned kernel, only prefill_flash rows are summed)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
OUT=$R/gpurun_out/pmc_pf_$TAG; mkdir -p $OUT; cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 tools/bench_prefill_attention.py "$@" > $OUT/p1.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -- python3 tools/bench_prefill_attention.py "$@" > $OUT/p2.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python3 tools/bench_prefill_attention.py "$@" > $OUT/p3.log 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for p in ("p1","p2","p3"):
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv",recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "prefill_flash" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(p,k,sum(v)/len(v),len(v))
PY
