#!/usr/bin/env python3
"""profiles/traffic.json (what bench.py's `roofline.traffic` and `binding` quote) from the rocprofv3 summaries under profiles/r02/:
per-launch means of FETCH_SIZE / WRITE_SIZE (KiB) and of the instruction counters, per kernel and batch."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = {  # key in traffic.json -> summary file, kernel the numbers are taken from
    "1024x10000@0.05s": ("profiles/r02/rocprof_band2_kernel_1024x10kb_score.txt", "wfa_band2_kernel<512, 3, 2, 1, false, true>"),
    "1024x10000@0.05c": ("profiles/r02/rocprof_band2_kernel_1024x10kb_cigar.txt", "wfa_band2_kernel<512, 3, 2, 1, true, true>"),
    "1250x50000@0.03s": ("profiles/r02/rocprof_generic_stream16_kernel_1250x50kb.txt", "wfa_batch_kernel<512, true, true, true>"),
}
out = {}
for key, (path, kern) in SRC.items():
    txt = open(os.path.join(ROOT, path)).read()
    pmc = {m.group(1): (float(m.group(2)), int(m.group(3))) for m in re.finditer(r"== pmc (\w+) = ([0-9.e+]+) per launch \((\d+) launches\) \[[^\]]*" + re.escape(kern.split("<")[0]), txt)}
    cells = int(re.search(r"cells/launch (\d+)", txt).group(1))
    n = pmc["FETCH_SIZE"][1]
    note = (f"{path}: separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over {kern.replace(', ', ',')}, means over {n} launches; "
            "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; "
            "the band2 kernel's reads are 8 bytes per lane, a width the guide calls uncalibrated, so this is an upper estimate; WRITE_SIZE as is: it matches the H stores)")
    out[key] = {
        "hbm_bytes_per_launch": (2 * pmc["FETCH_SIZE"][0] + pmc["WRITE_SIZE"][0]) * 1024,
        "fetch_size_kib": pmc["FETCH_SIZE"][0], "write_size_kib": pmc["WRITE_SIZE"][0],
        "valu_insts_per_launch": pmc["SQ_INSTS_VALU"][0], "salu_insts_per_launch": pmc["SQ_INSTS_SALU"][0], "lds_insts_per_launch": pmc["SQ_INSTS_LDS"][0],
        "cells_per_launch": cells, "source": note,
        "valu_source": f"{path}: rocprofv3 --pmc SQ_INSTS_VALU (wave-instructions), mean over {pmc['SQ_INSTS_VALU'][1]} launches",
    }
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if "source" not in kk} for k, v in out.items()}, indent=1))
