#!/bin/bash
# Development aid: compile mwf_band2.hip (dev subset of variants) to gfx950 assembly and print the register / spill table.
# Usage: profiles/asm_band2.sh [extra flags]   -> /tmp/band2_new.s, /tmp/n512.s (the 512x3 score-only 2-bit kernel)
cd "$(dirname "$0")/../miniwfa_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-const-variable -I ../../include -I . -DMWF_BAND_DEV "$@" --offload-device-only -S mwf_band2.hip -o /tmp/band2_new.s 2>&1 | grep -v hip-link
grep -E "sgpr_spill_count|vgpr_count|vgpr_spill_count|private_segment_fixed_size:|\.name:" /tmp/band2_new.s | paste - - - - - | sed 's/ \+/ /g' | sed 's/.*wfa_band2_kernelI//'
awk '/^_ZN3mwf12_GLOBAL__N_116wfa_band2_kernelILi512ELi3ELi2ELi1ELb0ELb1EEEvNS_9BatchArgsE:/{p=1} p{print} /^\.Lfunc_end/{if(p){exit}}' /tmp/band2_new.s > /tmp/n512.s
grep "amdhsa_group_segment_fixed_size" /tmp/band2_new.s | sort | uniq -c
