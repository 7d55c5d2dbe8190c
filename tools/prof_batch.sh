cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_b32 $R/gpurun_out/prof_b8
cd $R
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b32 -- python3 bench.py --batch 32 --input-len 1024 --steps 16 --warmup 2 --skip-cpu --skip-bf16 --skip-shapes --skip-ttft > gpurun_out/prof_b32.log 2>&1 && \
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b8 -- python3 bench.py --batch 8 --input-len 32 --steps 32 --warmup 2 --skip-cpu --skip-bf16 --skip-shapes --skip-ttft > gpurun_out/prof_b8.log 2>&1
tail -q -n 2 gpurun_out/prof_b32.log gpurun_out/prof_b8.log
