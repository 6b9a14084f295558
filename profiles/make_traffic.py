#!/usr/bin/env python3
"""profiles/traffic.json (what bench.py's `roofline.traffic` / candidates quote) from the rocprofv3 summaries under profiles/r05/:
per-launch means of FETCH_SIZE / WRITE_SIZE (KiB) and of the instruction counters, per kernel and workload."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from miniwfa_amd.build import kernel_fingerprint
RDIR = os.environ.get("MWF_PROFILE_DIR", "profiles/r06")   # the round's summaries


def git_state():
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "miniwfa_amd/csrc"], capture_output=True, text=True).stdout.strip())
        return head + ("+uncommitted kernel edits" if dirty else "")
    except Exception:
        return None


def fp_of(txt, kern):
    """The fingerprint the summary recorded on the GPU box for the kernel's source file (profiles/summarize.py); summaries older than that line: the
    tree make_traffic.py runs on (right only when run straight after the collection, before any kernel edit)."""
    m = re.search(r"== fingerprint " + re.escape(kern.split("<")[0]) + r" (\w+)", txt)
    return m.group(1) if m else kernel_fingerprint(kern)


# key in traffic.json -> (summary file, kernel name prefix(es) the numbers are taken from, read width in bytes per lane)
SRC = {
    # (round 5: the bench rotates four seeds through its steps — one runs on three chunk slots per wave, three on four: the launch-weighted mean of the two forms)
    "1024x10000@0.05s": (RDIR + "/rocprof_band2_kernel_1024x10kb_score.txt", ["wfa_band2_kernel<512, 3, 2, 1, false, true>", "wfa_band2_kernel<512, 4, 2, 1, false, true>"], 8),
    "1024x10000@0.05c": (RDIR + "/rocprof_band2_kernel_1024x10kb_cigar.txt", ["wfa_band2_kernel<512, 3, 2, 1, true, true>", "wfa_band2_kernel<512, 4, 2, 1, true, true>"], 8),
    "1250x50000@0.03s": (RDIR + "/rocprof_band2_span_kernel_1250x50kb.txt", ["wfa_band2_kernel<1024, 5, 2, 1, false, true>"], 8),
    "1250x50000@0.03s:generic16": (RDIR + "/rocprof_generic_stream16_kernel_1250x50kb.txt", ["wfa_batch_kernel<512, true, true, true, 0>"], 8),
    "c4_like_150kb:score": (RDIR + "/rocprof_sys_kernel_c4_score.txt", ["wfa_sys_kernel"], 4),
    "c4_like_150kb:cigar_highmem": (RDIR + "/rocprof_sys_kernel_c4_cigar.txt", ["wfa_sys_kernel"], 4),
    "c4_like_150kb:cigar_lowmem_p5000": (RDIR + "/rocprof_sys_kernel_c4_lowmem.txt", ["wfa_sys_kernel", "wfa_sys_seg_kernel"], 4),
    "mhc_like_5Mb:score": (RDIR + "/rocprof_sys_kernel_mhc_score.txt", ["wfa_sys_kernel"], 16),
    "mhc_like_5Mb:cigar_lowmem_p5000": (RDIR + "/rocprof_sys_kernel_mhc_lowmem.txt", ["wfa_sys_kernel", "wfa_sys_seg_kernel"], 16),
}
out = {}
for key, (path, kerns, width) in SRC.items():
    full = os.path.join(ROOT, path)
    if not os.path.exists(full):
        continue
    txt = open(full).read()
    pmc = {}
    for m in re.finditer(r"== pmc (\w+) = ([0-9.e+]+) per launch \((\d+) launches\) \[([^\]]*)", txt):
        name, val, n, kern = m.group(1), float(m.group(2)), int(m.group(3)), m.group(4)
        if any(k.split("<")[0] in kern and (("<" not in k) or k[:40] in kern or k.replace(", ", ",")[:30] in kern.replace(", ", ",")) for k in kerns):
            tot, cnt = pmc.get(name, (0.0, 0))
            if "wfa_band2_kernel" in kern and len(kerns) > 1:   # geometries of ONE class over the bench's rotation of seeds: launch-weighted mean
                pmc[name] = ((tot * cnt + val * n) / (cnt + n), cnt + n)
            else:
                pmc[name] = (tot + val, max(cnt, n))     # a mode that launches the kernel in two forms (low-memory passes): their sum per call
    if "FETCH_SIZE" not in pmc:
        continue
    forms = {m.group(4) for m in re.finditer(r"== pmc (FETCH_SIZE) = ([0-9.e+]+) per launch \((\d+) launches\) \[([^\]]*)", txt) if "wfa_sys_kernel" in m.group(4) or "wfa_sys_seg_kernel" in m.group(4)}
    if "lowmem" in key and len(forms) == 1:   # both passes of the low-memory mode ran the same kernel form: two launches per call
        pmc = {k: (v[0] * 2, v[1]) for k, v in pmc.items()}
    m = re.search(r"cells/launch (\d+)", txt)
    cells = int(m.group(1)) if m else None
    note = (f"{path}: separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over {', '.join(kerns)}, means per launch; "
            "bytes = 2048 x FETCH_SIZE + 1024 x WRITE_SIZE: the units measured on this device by profiles/micro/fetch_calib.hip "
            "(kernels that move exactly 1 GiB with 2-, 4-, 8- and 16-byte coalesced accesses per lane: 2048 bytes per FETCH_SIZE unit at every "
            "width, 1024 per WRITE_SIZE unit; profiles/r03/fetch_calibration.txt) — the doubling MI355X_MICROARCH.md prescribes; "
            f"this kernel's row reads are {width} bytes per lane")
    out[key] = {
        "hbm_bytes_per_launch": (2 * pmc["FETCH_SIZE"][0] + pmc["WRITE_SIZE"][0]) * 1024,
        "fetch_size_kib": pmc["FETCH_SIZE"][0], "write_size_kib": pmc["WRITE_SIZE"][0],
        "valu_insts_per_launch": pmc.get("SQ_INSTS_VALU", (None,))[0], "salu_insts_per_launch": pmc.get("SQ_INSTS_SALU", (None,))[0],
        "lds_insts_per_launch": pmc.get("SQ_INSTS_LDS", (None,))[0],
        "vmem_insts_per_launch": (pmc.get("SQ_INSTS_VMEM_RD", (0,))[0] or 0) + (pmc.get("SQ_INSTS_VMEM_WR", (0,))[0] or 0),
        "wait_any_over_wave_cycles": (pmc["SQ_WAIT_ANY"][0] / pmc["SQ_WAVE_CYCLES"][0]) if "SQ_WAIT_ANY" in pmc and "SQ_WAVE_CYCLES" in pmc else None,
        "source": note,
        "valu_source": f"{path}: rocprofv3 --pmc SQ_INSTS_VALU (wave-instructions), mean per launch",
        # what the profile is a profile OF: bench.py compares these with the tree and the kernel it timed (roofline.frac_stale)
        "kernels": kerns, "kernel_fingerprint": fp_of(txt, kerns[0]), "collected_at_commit": git_state(),
    }
    if cells:
        out[key]["cells_per_launch"] = cells
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if "source" not in kk} for k, v in out.items()}, indent=1))
