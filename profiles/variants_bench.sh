#!/bin/bash
# bench the headline batch (kernel ms) on every profiles/_<name>_libmwf_hip.so given.  Usage: profiles/variants_bench.sh "<bench args>" name...
cd "$(dirname "$0")/.."
ARGS=$1; shift
P='import json,sys; d=json.load(sys.stdin); print("%.3f ms kernel, %.3f ms/step, grid %d block %d retries %d" % (d["roofline"]["kernel_ms"], d["ms_per_step"], d["config"]["grid"], d["config"]["block"], d["n_retries"]))'
for v in "$@"; do
  echo -n "$v $ARGS: "; MWF_HIP_LIB=profiles/_${v}_libmwf_hip.so timeout 150 python bench.py --extras 0 --cpu-sample 0 --steps 5 $ARGS 2>&1 | tail -1 | python -c "$P"
done
