#!/bin/bash
# Round 5, second pass: the stall-side counters (FIFO-full, instruction-cycle and cache-busy counters) for one vs two workgroups per CU.
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/cores2; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
for W in 1 0; do
  TAG=$([ $W = 1 ] && echo one || echo two)
  N=$([ $W = 1 ] && echo 512 || echo 1024)
  D="$OUT/$TAG"; mkdir -p "$D"
  for PMC in "SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VSKIPPED SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
             "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT" \
             "SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL SQC_TC_REQ" \
             "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_SENDMSG SQ_INSTS SQ_ITEMS SQ_BUSY_CYCLES SQ_CYCLES" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    NAME=$(echo "$PMC" | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d "$D/pmc_$NAME" -o p -- python profiles/coresidency_probe.py $W $N > "$D/pmc_$NAME.out" 2> "$D/pmc_$NAME.err" || echo "pmc pass $NAME failed" >> "$D/errors.txt"
  done
  { python profiles/summarize.py "$D"; cat "$D/errors.txt" 2>/dev/null; } > "$OUT/$TAG.txt" 2>&1
  find "$D" -name "*.db" -delete
done
sed 's/\[void mwf::(anonymous namespace):://' "$OUT/one.txt" "$OUT/two.txt" | cut -c1-150
