#!/bin/bash
# Is the packed band kernel's loop body evicting itself from the instruction cache?  (a) co-residency: the headline batch with one and
# two workgroups per CU; (b) instruction-cache counters of the library given in MWF_HIP_LIB (default: the in-tree build).
# Usage: profiles/icache_probe.sh <tag>
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-x}
OUT=gpurun_out/icache_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
P='import json,sys; d=json.load(sys.stdin); print("%.3f ms kernel, grid %d block %d retries %d" % (d["roofline"]["kernel_ms"], d["config"]["grid"], d["config"]["block"], d["n_retries"]))'
for A in "--pairs 256" "--pairs 512" "--pairs 1024" "--pairs 1024 --slots-per-cu 1" "--pairs 1024 --cigar"; do
  echo -n "$A: "; timeout 200 python bench.py --extras 0 --cpu-sample 0 --steps 5 $A 2>&1 | tail -1 | python -c "$P"
done 2>&1 | tee "$OUT/coresidency.txt"
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*" | sort -u | tr '\n' ' ' > "$OUT/avail.txt"; cat "$OUT/avail.txt"; echo
BENCH="python bench.py --steps 3 --warmup 1 --cpu-sample 0 --extras 0"
for PMC in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  NAME=$(echo "$PMC" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $PMC -d "$OUT/pmc_$NAME" -o p -- $BENCH > /dev/null 2> "$OUT/pmc_$NAME.err" || { echo "pmc pass $NAME failed"; tail -3 "$OUT/pmc_$NAME.err"; }
done
python profiles/summarize.py "$OUT" 2>&1 | grep -v "^== kernel" | sed 's/\[void mwf::(anonymous namespace):://' | cut -c1-120 | tee "$OUT/pmc.txt"
find "$OUT" -name "*.db" -delete
