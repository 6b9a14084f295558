#!/bin/bash
# Instruction mix of the dominant kernel: rocprofv3 --pmc passes over a short bench run.  Usage: profiles/pmc_quick.sh <tag> [bench args]
cd "$(dirname "$0")/.." || exit 1
TAG=$1; shift
OUT=gpurun_out/pmcq_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python bench.py --steps 3 --warmup 1 --cpu-sample 0 --extras 0 $*"
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  NAME=$(echo "$PMC" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $PMC -d "$OUT/pmc_$NAME" -o p -- $BENCH > /dev/null 2> "$OUT/pmc_$NAME.err" || echo "pmc pass $NAME failed"
done
python profiles/summarize.py "$OUT" 2>&1 | grep -v "^== kernel" | sed 's/\[void mwf::(anonymous namespace):://' | cut -c1-110
