#!/bin/bash
# Every fuzzer of tests/fuzzlib.py on seeds the driver-run suite does not use (tests/test_gpu_fuzz.py fixes its own): a bug hunt, not a test.
# Usage (GPU box): bash profiles/fuzz_campaign.sh [first seed] [seeds]   -> gpurun_out/fuzz_campaign.log
cd "$(dirname "$0")/.." || exit 1
S0=${1:-21}; N=${2:-4}
LOG=gpurun_out/fuzz_campaign.log; mkdir -p gpurun_out; : > $LOG
for ((s = S0; s < S0 + N; ++s)); do
  for f in fuzz_fold fuzz_default_routing fuzz_all_kernels_oracle fuzz_band2_oracle fuzz_ring16 fuzz_seq2 fuzz_chain; do
    timeout 900 python profiles/$f.py $s 2>&1 | grep -E "BAD|FUZZ|FAILED|Error|error" >> $LOG
  done
done
grep -c "OK seed" $LOG; grep -E "FAILED|BAD|rror" $LOG | head -20
