#!/bin/bash
# Builds library variants for kernel experiments: tools/build_variants.sh name1 "flags1" name2 "flags2" ...
# Each variant recompiles csrc/w4a16.hip with its -D flags and links it with the tree's other objects into
# variants/libmi355q_<name>.so (tools/sweep_variants.sh runs the benchmark with each).
set -e
cd "$(dirname "$0")/../vllm-fork_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../variants /tmp/variants_obj
pids=()
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c w4a16.hip -o /tmp/variants_obj/w4a16_$name.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/libmi355q_$name.so /tmp/variants_obj/w4a16_$name.o \
      w8a16.o w8a8.o moe.o elementwise.o linear.o attention.o sampler.o allreduce.o capi.o && echo "built $name" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
