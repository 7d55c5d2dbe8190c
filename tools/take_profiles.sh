#!/bin/bash
# Round profiles of the headline workload (run on the GPU box from the repo root):
#   kernel-trace stats of a full bench run, then FETCH_SIZE and WRITE_SIZE in passes of their own,
#   folded into the record bench.py reads for roofline.traffic.  Results land in gpurun_out/$1/.
set -e
OUT=gpurun_out/${1:-profiles}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 64 --warmup 8 --skip-cpu --skip-bf16 --skip-shapes --skip-configs > $OUT/stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --steps 8 --warmup 2 --skip-cpu --skip-bf16 --skip-shapes --skip-configs --skip-ttft > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --steps 8 --warmup 2 --skip-cpu --skip-bf16 --skip-shapes --skip-configs --skip-ttft > $OUT/write.log 2>&1
python tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/decode_8b_gptq_pmc_fetch_write.json
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -n 1) $OUT/decode_8b_gptq_kernel_stats.csv
tail -n 1 $OUT/stats.log | cut -c 1-400
