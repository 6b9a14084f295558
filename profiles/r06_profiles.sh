#!/bin/bash
# Round-6 rocprofv3 evidence, run on the GPU box: kernel-trace stats + separate PMC passes (never combined with sys-trace) for
#   the packed band kernel (1024 x 10 kb, score and CIGAR), its 1024-thread span geometry (1250 x 50 kb), the lane kernel (40 000 x 150 bp),
#   the mid kernel (one 2 kb pair, 256 x 2 kb) and the whole-device kernel (C4-like 150 kb pair and MHC-like 5 Mb pair, every mode).
# Summaries land in gpurun_out/prof_* and gpurun_out/pmcc_*; copy them to profiles/r06/ (profiles/r06_collect.sh), which regenerates
# profiles/traffic.json (profiles/make_traffic.py) with the fingerprints of the kernel sources the counters were collected on.
# Usage: bash profiles/r06_profiles.sh [band|sys|small|all]
cd "$(dirname "$0")/.." || exit 1
WHAT=${1:-all}
if [ "$WHAT" = all ] || [ "$WHAT" = band ]; then
  bash profiles/run_profile.sh r06_band2_score > /dev/null 2>&1
  bash profiles/run_profile.sh r06_band2_cigar --cigar > /dev/null 2>&1
  bash profiles/run_profile.sh r06_span --config 5 --pairs 1250 --steps 2 > /dev/null 2>&1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = small ]; then
  bash profiles/pmc_cmd.sh r06_lane python profiles/kernel_once.py lane > /dev/null 2>&1
  bash profiles/pmc_cmd.sh r06_mid1 python profiles/kernel_once.py mid1 > /dev/null 2>&1
  bash profiles/pmc_cmd.sh r06_mid256 python profiles/kernel_once.py mid256 > /dev/null 2>&1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = sys ]; then
  for pm in "c4 score" "c4 cigar" "c4 lowmem" "mhc score" "mhc lowmem"; do
    set -- $pm
    bash profiles/pmc_cmd.sh r06_sys_$1_$2 python profiles/sys_modes.py $1 $2 > /dev/null 2>&1
  done
fi
for t in prof_r06_band2_score prof_r06_band2_cigar prof_r06_span pmcc_r06_lane pmcc_r06_mid1 pmcc_r06_mid256 pmcc_r06_sys_c4_score pmcc_r06_sys_c4_cigar pmcc_r06_sys_c4_lowmem pmcc_r06_sys_mhc_score pmcc_r06_sys_mhc_lowmem; do
  [ -f gpurun_out/$t/summary.txt ] && { echo "=== $t"; head -12 gpurun_out/$t/summary.txt | cut -c1-170; }
done
