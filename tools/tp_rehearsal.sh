#!/bin/bash
# Rehearsal of the multi-GPU bench on a ONE-GPU box: the same bench.py command lines the driver uses for
# N = 2, 4 (and 8: the TP degree the 70B configuration is quoted on), with all ranks sharing GPU 0 and the collectives going through gloo (RCCL needs one device
# per rank).  Checks that the TP = N model decodes the same greedy tokens as TP = 1 (same synthetic
# checkpoint, sharded by the parameter loaders).  usage: tools/tp_rehearsal.sh OUTDIR [model] ["2 4 8"]
set -e
OUT=$1; MODEL=${2:-llama3-70b}; RANKS=${3:-"2 4"}
mkdir -p $OUT
COMMON="--model $MODEL --steps 12 --warmup 2 --skip-cpu --skip-bf16 --skip-shapes"
timeout -k 10 400 python bench.py --gpus 1 $COMMON --tokens-out $OUT/tokens_tp1.json > $OUT/bench_tp1.json 2> $OUT/bench_tp1.err
for N in $RANKS; do
  MI355Q_DIST_BACKEND=gloo MI355Q_SHARE_GPU=1 timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
    --master-addr 127.0.0.1 --master-port $((29510 + N)) bench.py --gpus $N $COMMON --tokens-out $OUT/tokens_tp$N.json \
    > $OUT/bench_tp$N.json 2> $OUT/bench_tp$N.err
done
python - $OUT $RANKS <<'PY'
import json, sys
out = sys.argv[1]
ref = json.load(open(f"{out}/tokens_tp1.json"))
ok = True
for n in [int(a) for a in sys.argv[2:]]:
    got = json.load(open(f"{out}/tokens_tp{n}.json"))
    same = 0
    for a, b in zip(ref, got):
        if a != b:
            break
        same += 1
    line = json.loads(open(f"{out}/bench_tp{n}.json").read().strip().splitlines()[-1])
    print(f"TP={n}: {same}/{len(ref)} leading decode steps equal to TP=1; first tokens {got[0]} vs {ref[0]}; "
          f"tok/s {line['value']:.1f} graph={line['graph']} scaling={line['scaling']} n_gpus={line['n_gpus']} "
          f"exchange: {line.get('tp_exchange')} {line.get('peer_allreduce_us_16k')}")
    # partial sums are rounded per rank before the all-reduce, so a near-tie may flip late in the sequence;
    # the first eight steps must agree
    ok &= same >= 8
print("REHEARSAL", "OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
PY
