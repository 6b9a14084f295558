#!/bin/bash
# Development loop on the GPU box: band-kernel parity subset, CIGAR/score timings, per-phase cycle counts.
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${1:-band or golden or full}" 2>&1 | tail -5
timeout 120 python profiles/cigar_pack_probe.py 2>&1 | tail -3
timeout 100 python bench.py --steps 5 --warmup 2 2>/dev/null > /tmp/b.json; python -c "import json; d=json.load(open(\"/tmp/b.json\")); print(\"bench\", d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"kernel_ms\"])"
if [ -f profiles/_timing_libmwf_hip.so ]; then
	MWF_HIP_LIB=profiles/_timing_libmwf_hip.so timeout 200 python profiles/timing_probe.py 2>&1 | tail -8 | cut -c 1-20,60-400
fi
