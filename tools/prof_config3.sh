# Kernel-trace stats of the AWQ + fp8-KV decode (BASELINE configuration 3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/prof_c3; cd $R
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -- python3 bench.py --quant awq --kv-dtype fp8 --steps 32 --warmup 4 --skip-cpu --skip-bf16 --skip-shapes > gpurun_out/prof_c3.log 2>&1
tail -1 gpurun_out/prof_c3.log | cut -c1-330
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_c3/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} {r['Percentage']}")
PY
