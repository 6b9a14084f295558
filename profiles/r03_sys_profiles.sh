#!/bin/bash
# Round-3 rocprofv3 evidence for the systolic whole-device kernel: kernel stats + PMC per long-pair mode -> gpurun_out/pmcc_sys_*/summary.txt
cd "$(dirname "$0")/.." || exit 1
for pm in "c4 score" "c4 cigar" "c4 lowmem" "mhc score" "mhc lowmem"; do
  set -- $pm
  bash profiles/pmc_cmd.sh sys_$1_$2 python profiles/sys_modes.py $1 $2 > /dev/null 2>&1
  echo "=== $1 $2"; grep -E "^[a-z0-9]+ (score|cigar|lowmem):|wfa_sys|wfa_coop|sys_walk|sys_finish" gpurun_out/pmcc_sys_$1_$2/summary.txt | cut -c1-170 | head -40
done
