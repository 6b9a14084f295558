# The measurements behind profiles/r02 (second half of round 2), one gpurun call: GPU tests, smoke, default bench line, rocprofv3 trace + PMC of the
# headline kernel (score, CIGAR), short-read batches, microbenchmarks, phase cycles of the packed and the balanced band kernel.
set -x
mkdir -p gpurun_out/r02b
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02b/gputest.log
python __graft_entry__.py --smoke > gpurun_out/r02b/smoke.log 2>&1
python bench.py > gpurun_out/r02b/bench_default.json 2> gpurun_out/r02b/bench_default.err
bash profiles/run_profile.sh b2s > /dev/null 2>&1; cp gpurun_out/prof_b2s/summary.txt gpurun_out/r02b/rocprof_band2_kernel_1024x10kb_score.txt
bash profiles/run_profile.sh b2c --cigar > /dev/null 2>&1; cp gpurun_out/prof_b2c/summary.txt gpurun_out/r02b/rocprof_band2_kernel_1024x10kb_cigar.txt
bash profiles/short_reads.sh > gpurun_out/r02b/short_reads.txt 2>&1
./profiles/micro/lds_rates > gpurun_out/r02b/lds_issue_rates_microbench.txt 2>&1
[ -f profiles/_b2t_libmwf_hip.so ] && MWF_HIP_LIB=profiles/_b2t_libmwf_hip.so python profiles/timing_probe_band2.py > gpurun_out/r02b/band2_phase_cycles.txt 2>&1
[ -f profiles/_b3t_libmwf_hip.so ] && for b in 512 1024; do echo "band3 block $b"; MWF_BAND3_BLOCK=$b MWF_HIP_LIB=profiles/_b3t_libmwf_hip.so python profiles/timing_probe_band3.py; done > gpurun_out/r02b/band3_phase_cycles.txt 2>&1
python profiles/band3_check.py > gpurun_out/r02b/band3_check.txt 2>&1
tail -3 gpurun_out/r02b/gputest.log; cat gpurun_out/r02b/smoke.log | tail -2; head -c 300 gpurun_out/r02b/bench_default.json
