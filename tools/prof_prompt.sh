# Kernel-trace stats of the prompt pass at a given length: bash tools/prof_prompt.sh LEN
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=${1:-2048}
mkdir -p $R/gpurun_out/prof_p$L; cd $R
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_p$L -- python3 bench.py --input-len $L --steps 2 --warmup 1 --skip-cpu --skip-bf16 --skip-shapes > gpurun_out/prof_p$L.log 2>&1
python3 - $L <<'PY'
import csv,glob,sys
f=glob.glob(f"gpurun_out/prof_p{sys.argv[1]}/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:9.2f} ms {r['Percentage']}")
PY
