#!/bin/bash
# Decode step time of Llama-3-8B GPTQ with every library build under variants/ (kernel experiments).
# usage: tools/sweep_variants.sh OUTDIR [extra bench flags]
OUT=$1; shift
mkdir -p $OUT
for so in variants/libmi355q_*.so; do
  name=$(basename $so .so); name=${name#libmi355q_}
  MI355Q_LIB_PATH=$PWD/$so timeout -k 10 240 python bench.py --skip-cpu --skip-shapes --skip-bf16 --skip-ttft "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -3 $OUT/$name.err; exit 1; }
  python - "$name" "$OUT/$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
r = d["roofline"]
print(f"{sys.argv[1]:10s} {d['value']:7.1f} tok/s  {d['ms_per_step']*1000:7.1f} us/step  gate_up {r['avg_launch_us']:.2f} us (min {r['min_launch_us']:.2f})  tokens {d['last_token_ids']}")
PY
done
