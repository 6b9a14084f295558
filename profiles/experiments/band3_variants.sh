#!/bin/bash
# kernel time of the balanced band kernel (band3 = 1 via MWF_BAND3=1) by threads per workgroup.  Usage: profiles/band3_variants.sh
cd "$(dirname "$0")/.."
P='import json,sys; d=json.load(sys.stdin); print("%.3f ms kernel, grid %d block %d retries %d" % (d["roofline"]["kernel_ms"], d["config"]["grid"], d["config"]["block"], d["n_retries"]))'
for lib in ""; do for blk in 512 768 1024; do for A in "--pairs 256" "--pairs 1024" "--pairs 1024 --cigar"; do
  echo -n "lib=$lib block=$blk $A: "; MWF_BAND3=1 MWF_BAND3_BLOCK=$blk timeout 200 python bench.py --extras 0 --cpu-sample 0 --steps 5 $A 2>&1 | tail -1 | python -c "$P"
done; done; done
