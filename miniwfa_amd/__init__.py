"""miniwfa_amd — MI355X (gfx950) implementation of lh3/miniwfa's exact WFA score-and-CIGAR path.

The product is the C-ABI shared library ``miniwfa_amd/csrc/libmwf_hip.so`` (headers in
``include/``); this package is the thin Python host side over it (``miniwfa_amd.api``) plus the
synthetic-pair generator the tests and the benchmark share (``miniwfa_amd.synth``).
"""
from .api import (MWF_F_CIGAR, MWF_F_DEBUG, MWF_F_NO_KALLOC, Batch, Engine, MwfOpt, MwfRst, cigar2score, cigar_str,
                  lib, opt_init, wfa_auto, wfa_batch, wfa_batch_multi, wfa_chain, wfa_chain_batch, wfa_auto_batch, wfa_exact, wfa_submit, async_stats, Job)

__all__ = ["MWF_F_CIGAR", "MWF_F_DEBUG", "MWF_F_NO_KALLOC", "Batch", "Engine", "MwfOpt", "MwfRst", "cigar2score",
           "cigar_str", "lib", "opt_init", "wfa_auto", "wfa_batch", "wfa_batch_multi", "wfa_chain", "wfa_chain_batch", "wfa_auto_batch", "wfa_exact", "wfa_submit", "async_stats", "Job"]
