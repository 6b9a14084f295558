"""tools/test-mwf: the reference's command line (main.c:19-92).  The reader is checked on CPU (FASTA, multi-line FASTA,
FASTQ, gzip); the full tool is checked on the GPU against the reference's own t3 answer and output format."""
import gzip
import os
import subprocess

import pytest

from miniwfa_amd import build as b
from conftest import load_golden, golden_inputs


@pytest.fixture(scope="module")
def cli():
    b.build()
    return b.build_cli()


def _write(tmp_path, name, text, gz=False):
    p = tmp_path / name
    if gz:
        with gzip.open(p, "wt") as f:
            f.write(text)
    else:
        p.write_text(text)
    return str(p)


def test_reader_formats(cli, tmp_path):
    fa1 = _write(tmp_path, "a.fa", ">s1 desc\nACGT\nACG\n>s2\nTTTT\n")
    fq2 = _write(tmp_path, "b.fq", "@r1\nACGTA\n+\nIIIII\n@r2 x\nGG\n+r2\n!!\n")
    out = subprocess.run([cli, fa1, fq2], env={**os.environ, "MWF_CLI_PARSE_ONLY": "1"}, capture_output=True, text=True, check=True).stdout
    assert out.splitlines() == ["s1\t7\tr1\t5", "s2\t4\tr2\t2"]
    gz1 = _write(tmp_path, "a.fa.gz", ">g\nAC\nGT\n", gz=True)
    out = subprocess.run([cli, gz1, fa1], env={**os.environ, "MWF_CLI_PARSE_ONLY": "1"}, capture_output=True, text=True, check=True).stdout
    assert out.splitlines() == ["g\t4\ts1\t7"]
    assert subprocess.run([cli], capture_output=True, text=True).returncode == 1   # usage


@pytest.mark.gpu
def test_t3_output_matches_reference_format(cli, tmp_path):
    t, q = golden_inputs(load_golden("exact_small.jsonl")[0])
    f1 = _write(tmp_path, "t3-0.fa", ">1\n" + t.decode() + "\n")
    f2 = _write(tmp_path, "t3-1.fa", ">2\n" + q.decode() + "\n")
    exp = "1\t61\t0\t61\t+\t2\t189\t0\t189\t155"
    for flags, tail in ((["-c"], "\t1X16=1X14=128I4=1X24="), ([], ""), (["-cp5"], "\t1X16=1X14=128I4=1X24="), (["-ct"], "\t1X16=1X14=128I4=1X24="),
                        (["-cu"], "\t1X16=1X18=128I1X24="), (["-ca"], None), (["-ce"], None)):
        r = subprocess.run([cli, *flags, f1, f2], capture_output=True, text=True, check=True)
        line = r.stdout.strip()
        if tail is not None:
            assert line == exp + tail, (flags, line)
        assert r.stderr.startswith("T\t1\t2\t") or "T\t1\t2\t" in r.stderr
    a = subprocess.run([cli, "-ca", f1, f2], capture_output=True, text=True, check=True).stdout.split("\t")
    assert a[9] == "272" and a[10].strip() == "1X16=1X18=118I1=10I24="     # SURVEY Appendix B, -a
    e = subprocess.run([cli, "-ce", f1, f2], capture_output=True, text=True, check=True).stdout.split("\t")
    assert e[9] == "128"


def test_reference_main_compiles_and_links_against_this_library(cli):
    """The drop-in claim with the reference's OWN caller: /root/reference/main.c (main.c:19-92), unchanged and not copied, compiles
    against include/miniwfa.h + include/kalloc.h and links with libmwf_hip.so (build container only: the sources do not travel)."""
    if not os.path.exists("/root/reference/main.c"):
        pytest.skip("reference sources not present (GPU box): the prebuilt binary is used by the GPU test below")
    from oracle.build_ref_main import build_ref_main
    exe = build_ref_main()
    if not exe:
        pytest.skip("the reference's main.c could not be built here (gcc / zlib missing)")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage: test-mwf" in r.stderr      # main.c:46-57
    needed = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libmwf_hip.so" in needed and "libmwf_ref" not in needed  # every mwf_* / kalloc symbol comes from the HIP library


@pytest.mark.gpu
def test_reference_main_on_this_library_gives_the_reference_output(cli, tmp_path):
    """... and run on the GPU it prints what tools/test-mwf prints and what the reference itself prints for its t3 fixture."""
    from oracle.build_ref_main import build_ref_main
    exe = build_ref_main()
    if not exe or not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref-main-on-libmwf_hip was not built (no reference sources where build() ran)")
    t, q = golden_inputs(load_golden("exact_small.jsonl")[0])
    f1 = _write(tmp_path, "t3-0.fa", ">1\n" + t.decode() + "\n")
    f2 = _write(tmp_path, "t3-1.fa", ">2\n" + q.decode() + "\n")
    exp = "1\t61\t0\t61\t+\t2\t189\t0\t189\t155"
    for flags, tail in ((["-c"], "\t1X16=1X14=128I4=1X24="), ([], ""), (["-cp5"], "\t1X16=1X14=128I4=1X24="), (["-cu"], "\t1X16=1X18=128I1X24="),
                        (["-ca"], None), (["-ce"], None), (["-cK"], "\t1X16=1X14=128I4=1X24=")):
        ours = subprocess.run([cli, *flags, f1, f2], capture_output=True, text=True, check=True).stdout
        theirs = subprocess.run([exe, *flags, f1, f2], capture_output=True, text=True, check=True).stdout
        assert ours == theirs, (flags, ours, theirs)
        if tail is not None:
            assert theirs.strip() == exp + tail, (flags, theirs)
