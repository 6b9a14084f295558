#!/bin/bash
# Instruction mix / wait cycles / memory traffic of the kernels an arbitrary command launches: rocprofv3 --kernel-trace --stats,
# then separate --pmc passes.  Usage: profiles/pmc_cmd.sh <tag> <command...>      -> gpurun_out/pmcc_<tag>/summary.txt
cd "$(dirname "$0")/.." || exit 1
TAG=$1; shift
OUT=gpurun_out/pmcc_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- "$@" > "$OUT/cmd_trace.out" 2> "$OUT/trace.err"
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  NAME=$(echo "$PMC" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $PMC -d "$OUT/pmc_$NAME" -o p -- "$@" > /dev/null 2> "$OUT/pmc_$NAME.err" || echo "pmc pass $NAME failed"
done
{ echo "# $*"; cat "$OUT/cmd_trace.out" | tail -5; python profiles/summarize.py "$OUT"; } > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete   # the sqlite files are large; the summary is what is kept
sed 's/\[void mwf::(anonymous namespace):://' "$OUT/summary.txt" | cut -c1-160
