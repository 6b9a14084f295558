"""bench.py's N > 1 start-up on CPU: a bare `python bench.py --gpus 2` (no WORLD_SIZE) must re-execute itself under
torch.distributed.run, and the two ranks must deal the pairs, gather every record with ONE all_gather and print ONE JSON line.
MWF_BENCH_BACKEND=gloo makes that a dry run on CPU with the oracle standing in for the GPU (the partition and the gather are the
same code on RCCL); the line is marked "dry_run": true and is not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


def test_launcher_command_line():
    import bench
    cmd = bench.launcher_argv(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29517)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]      # the ranks get the caller's arguments unchanged
    p = bench.free_port()
    assert 1024 < p < 65536


@pytest.mark.parametrize("extra", [[], ["--config", "5"]])
def test_bare_two_rank_launch_dry_run_on_gloo(extra):
    r = _run(["--gpus", "2", "--steps", "2"] + extra, {"MWF_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE line
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["dry_run"] is True
    assert d["gathered_records_match_oracle"] is True     # every pair's record reached rank 0, in global order
    assert d["scaling"] == ("strong" if extra else "weak")
    if extra:   # configs[4]'s shape: a fixed total dealt over the ranks
        assert d["config"]["pairs_total"] == 40 and d["config"]["pairs_this_rank"] == 20
    else:       # configs[2]'s: a fixed share per rank
        assert d["config"]["pairs_total"] == 2 * d["config"]["pairs_this_rank"]


def test_launcher_and_gpus_must_agree():
    r = _run(["--gpus", "2", "--steps", "1"], {"MWF_BENCH_BACKEND": "gloo", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_single_rank_dry_run_and_no_gpu_message():
    r = _run(["--steps", "1", "--warmup", "0"], {"MWF_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_lines(r.stdout)[0]
    assert d["n_gpus"] == 1 and d["dry_run"] is True and d["gathered_records_match_oracle"] is True
    import torch
    if not torch.cuda.is_available():   # the real path refuses to run without a GPU: no CPU fallback
        r = _run(["--steps", "1", "--warmup", "0", "--extras", "0", "--cpu-sample", "0"])
        assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)


def test_bench_reads_the_real_pairs_when_they_are_dropped_in(tmp_path):
    """bench.py --c4 A B / --mhc A B (the reference's Zenodo pairs, README.md:82-88): the FASTA / FASTQ(.gz) reader takes the first record's bytes as they are."""
    import gzip
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fa = tmp_path / "a.fa"
    fa.write_bytes(b">one desc\nACGT\nacgtN\n>two\nTTTT\n")
    fq = tmp_path / "b.fq.gz"
    with gzip.open(fq, "wb") as f:
        f.write(b"@r1\nGATTACA\n+\nIIIIIII\n@r2\nCC\n+\nII\n")
    assert bench.read_first_fasta(str(fa)) == b"ACGTacgtN"
    assert bench.read_first_fasta(str(fq)) == b"GATTACA"
    assert bench.REAL_PAIRS["c4_like_150kb"][1] == 26917 and bench.REAL_PAIRS["mhc_like_5Mb"][1] == 229868
