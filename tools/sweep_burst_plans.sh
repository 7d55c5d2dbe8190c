#!/bin/bash
# Per-kernel time of the four decode GEMVs under forced (waves, ring depth) plans of the burst kernel.
# usage: tools/sweep_burst_plans.sh  (prints one line per plan)
run() { echo -n "ns=$1 nw=$2 cpw=$3: "; MI355Q_BURST_NS=$1 MI355Q_BURST_NW=$2 MI355Q_BURST_CPW=$3 timeout -k 10 200 python tools/bench_layer_kernels.py 2>&1 | grep -v amdgpu.ids; }
echo -n "default: "; timeout -k 10 200 python tools/bench_layer_kernels.py 2>&1 | grep -v amdgpu.ids
for plan in "7 16 1" "7 8 1" "7 4 2" "7 8 2"; do run $plan; done          # gate_up: 32 chunks
for plan in "2 16 1" "2 8 2" "2 8 4" "2 4 4" "2 16 2"; do run $plan; done # qkv: 32 chunks
for plan in "1 16 7" "1 8 7" "1 14 2" "1 7 4" "1 8 2" "1 14 4" "1 16 2" "1 8 4" "1 4 4"; do run $plan; done  # down: 112 chunks, o: 32
echo -n "default: "; timeout -k 10 200 python tools/bench_layer_kernels.py 2>&1 | grep -v amdgpu.ids
