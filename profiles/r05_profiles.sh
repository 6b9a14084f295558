#!/bin/bash
# Round-5 rocprofv3 evidence, run on the GPU box: kernel-trace stats + separate PMC passes for
#   the packed band kernel (1024 x 10 kb, score and CIGAR), its 1024-thread span geometry (1250 x 50 kb), the lane kernel (40 000 x 150 bp),
#   the mid kernel (one 2 kb pair, 256 x 2 kb) and the whole-device kernel (C4-like 150 kb pair and MHC-like 5 Mb pair, every mode).
# Summaries land in gpurun_out/prof_* and gpurun_out/pmcc_*; copy them to profiles/r05/ (profiles/r05_collect.sh) and regenerate
# profiles/traffic.json with profiles/make_traffic.py.
cd "$(dirname "$0")/.." || exit 1
bash profiles/run_profile.sh r05_band2_score > /dev/null 2>&1
bash profiles/run_profile.sh r05_band2_cigar --cigar > /dev/null 2>&1
bash profiles/run_profile.sh r05_span --config 5 --pairs 1250 --steps 2 > /dev/null 2>&1   # (the packed kernel's span geometry; the generic 16-bit-ring kernel's summary of the same batch dates from before it)
bash profiles/pmc_cmd.sh r05_lane python profiles/kernel_once.py lane > /dev/null 2>&1
bash profiles/pmc_cmd.sh r05_mid1 python profiles/kernel_once.py mid1 > /dev/null 2>&1
bash profiles/pmc_cmd.sh r05_mid256 python profiles/kernel_once.py mid256 > /dev/null 2>&1
for pm in "c4 score" "c4 cigar" "c4 lowmem" "mhc score" "mhc lowmem"; do
  set -- $pm
  bash profiles/pmc_cmd.sh r05_sys_$1_$2 python profiles/sys_modes.py $1 $2 > /dev/null 2>&1
done
for t in prof_r05_band2_score prof_r05_band2_cigar prof_r05_span pmcc_r05_lane pmcc_r05_mid1 pmcc_r05_mid256 pmcc_r05_sys_c4_score pmcc_r05_sys_c4_cigar pmcc_r05_sys_c4_lowmem pmcc_r05_sys_mhc_score pmcc_r05_sys_mhc_lowmem; do
  echo "=== $t"; head -12 gpurun_out/$t/summary.txt | cut -c1-170
done
