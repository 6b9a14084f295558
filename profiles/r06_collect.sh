#!/bin/bash
# Copy the summaries profiles/r06_profiles.sh left under gpurun_out/ into profiles/r06/ (tracked) and regenerate profiles/traffic.json.
cd "$(dirname "$0")/.." || exit 1
mkdir -p profiles/r06
cpy() { [ -f "gpurun_out/$1/summary.txt" ] && cp "gpurun_out/$1/summary.txt" "profiles/r06/$2" && echo "profiles/r06/$2"; }
cpy prof_r06_band2_score rocprof_band2_kernel_1024x10kb_score.txt
cpy prof_r06_band2_cigar rocprof_band2_kernel_1024x10kb_cigar.txt
cpy prof_r06_span rocprof_band2_span_kernel_1250x50kb.txt
cpy pmcc_r06_lane rocprof_lane_kernel_40000x150bp.txt
cpy pmcc_r06_mid1 rocprof_mid_kernel_1x2kb.txt
cpy pmcc_r06_mid256 rocprof_mid_kernel_256x2kb.txt
for pm in c4_score c4_cigar c4_lowmem mhc_score mhc_lowmem; do cpy pmcc_r06_sys_$pm rocprof_sys_kernel_$pm.txt; done
MWF_PROFILE_DIR=profiles/r06 python profiles/make_traffic.py > /dev/null && echo profiles/traffic.json
