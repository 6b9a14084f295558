#!/bin/bash
# Round 5: what do two co-resident workgroups of the packed band kernel contend for?  One PMC pass per counter group for
# one vs two workgroups per CU on the headline batch.  -> gpurun_out/cores/{one,two}.txt (+ counters.txt: what rocprofv3 lists)
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/cores; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_all.txt" 2>&1
grep -o "SQ[C]*_[A-Z0-9_]*" "$OUT/counters_all.txt" | sort -u > "$OUT/counters.txt"
for W in 1 0; do
  TAG=$([ $W = 1 ] && echo one || echo two)
  N=$([ $W = 1 ] && echo 512 || echo 1024)   # the same number of rounds (two) per workgroup slot
  D="$OUT/$TAG"; mkdir -p "$D"
  python profiles/coresidency_probe.py $W $N > "$D/plain.out" 2>&1
  for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
             "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_SMEM" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
             "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_INPUT_VALID_READYB" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
             "SQ_CYCLES SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES" ; do
    NAME=$(echo "$PMC" | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d "$D/pmc_$NAME" -o p -- python profiles/coresidency_probe.py $W $N > "$D/pmc_$NAME.out" 2> "$D/pmc_$NAME.err" || echo "pmc pass $NAME failed" >> "$D/errors.txt"
  done
  { cat "$D/plain.out"; python profiles/summarize.py "$D"; cat "$D/errors.txt" 2>/dev/null; } > "$OUT/$TAG.txt" 2>&1
  find "$D" -name "*.db" -delete
done
sed 's/\[void mwf::(anonymous namespace):://' "$OUT/one.txt" "$OUT/two.txt" | cut -c1-150
