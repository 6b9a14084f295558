#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats plus separate PMC passes for a bench workload.
# Usage: profiles/run_profile.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-r02}; shift || true
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python bench.py --steps 3 --warmup 1 --cpu-sample 0 --extras 0 $*"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
# PMC passes, each alone with kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE 3 TCC slots, WRITE_SIZE 2)
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  NAME=$(echo "$PMC" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $PMC -d "$OUT/pmc_$NAME" -o p -- $BENCH > /dev/null 2> "$OUT/pmc_$NAME.err" || echo "pmc pass $NAME failed" >> "$OUT/errors.txt"
done
{ echo "# $BENCH"; tail -1 "$OUT/bench_trace.json" | python -c "import json,sys; d=json.load(sys.stdin); print('# bench line of the traced run: value %.4f Gbp/s, %.3f ms/step, kernel %.3f ms, cells/launch %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['cells_per_launch']))"; python profiles/summarize.py "$OUT"; } > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete   # the sqlite files are large; the summary is what is kept
cat "$OUT/summary.txt"
