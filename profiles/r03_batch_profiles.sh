#!/bin/bash
# Round-3 rocprofv3 evidence for the one-workgroup-per-pair kernels: run on the GPU box; summaries land in gpurun_out/prof_*/summary.txt
cd "$(dirname "$0")/.." || exit 1
bash profiles/run_profile.sh r03_band2_score > /dev/null 2>&1
bash profiles/run_profile.sh r03_band2_cigar --cigar > /dev/null 2>&1
bash profiles/run_profile.sh r03_generic16 --config 5 --pairs 1250 --steps 2 > /dev/null 2>&1
for t in r03_band2_score r03_band2_cigar r03_generic16; do echo "=== $t"; head -40 gpurun_out/prof_$t/summary.txt; done
