#!/bin/bash
# Batches of shorter pairs, score-only, 5 % divergence.
cd "$(dirname "$0")/.."
P='import json,sys; d=json.load(sys.stdin); print("%-22s %.3f Gbp/s  kernel %.3f ms  block %d grid %d retries %d" % (sys.argv[1], d["kernel_gbps"], d["roofline"]["kernel_ms"], d["config"]["block"], d["config"]["grid"], d["n_retries"]))'
for spec in "40000 150" "20000 300" "20000 1000" "8192 3000" "4096 5000"; do
  set -- $spec
  timeout 300 python bench.py --extras 0 --cpu-sample 0 --steps 5 --pairs $1 --len $2 $EXTRA 2>/dev/null | tail -1 | python -c "$P" "$1 x $2 bp"
done
