set -x
mkdir -p gpurun_out/r02b
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02b/gputest.log
python bench.py > gpurun_out/r02b/bench_default.json 2> gpurun_out/r02b/bench_default.err
bash profiles/run_profile.sh b2s > /dev/null 2>&1; cp gpurun_out/prof_b2s/summary.txt gpurun_out/r02b/rocprof_band2_kernel_1024x10kb_score.txt
bash profiles/run_profile.sh b2c --cigar > /dev/null 2>&1; cp gpurun_out/prof_b2c/summary.txt gpurun_out/r02b/rocprof_band2_kernel_1024x10kb_cigar.txt
bash profiles/short_reads.sh > gpurun_out/r02b/short_reads.txt 2>&1
./profiles/micro/lds_rates > gpurun_out/r02b/lds_issue_rates_microbench.txt 2>&1
for b in 512 1024; do echo "band3 block $b"; MWF_BAND3_BLOCK=$b MWF_HIP_LIB=profiles/_b3t_libmwf_hip.so python profiles/timing_probe_band3.py; done > gpurun_out/r02b/band3_phase_cycles.txt 2>&1
python profiles/band3_check.py > gpurun_out/r02b/band3_check.txt 2>&1
python profiles/call_latency.py > gpurun_out/r02b/call_latency.txt 2>&1
tail -3 gpurun_out/r02b/gputest.log; head -c 400 gpurun_out/r02b/bench_default.json
