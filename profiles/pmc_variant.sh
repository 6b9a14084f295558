#!/bin/bash
# PMC passes (L2 hit rate, fetch size, read latency, issue stalls) of the headline batch on a variant library.  Usage: profiles/pmc_variant.sh <variant name> [per_cu] [pairs]
cd "$(dirname "$0")/.." || exit 1
V=$1; W=${2:-0}; N=${3:-1024}
OUT=gpurun_out/pv_$V; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp MWF_HIP_LIB=profiles/_${V}_libmwf_hip.so
python profiles/coresidency_probe.py $W $N > "$OUT/plain.out" 2>&1
for PMC in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  NAME=$(echo "$PMC" | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d "$OUT/pmc_$NAME" -o p -- python profiles/coresidency_probe.py $W $N > /dev/null 2> "$OUT/pmc_$NAME.err" || echo "pmc pass $NAME failed"
done
{ cat "$OUT/plain.out"; python profiles/summarize.py "$OUT"; } 2>&1 | sed 's/\[void mwf::(anonymous namespace):://' | cut -c1-110 > "$OUT.txt"
rm -rf "$OUT"; cat "$OUT.txt"
