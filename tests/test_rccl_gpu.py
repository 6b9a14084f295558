"""The multi-GPU result gather on real hardware: torch.distributed backend "nccl" (= RCCL on ROCm), device tensors.

The box the GPU tests run on has one GPU, so this is world_size 1 — every RCCL call of the N > 1 path (communicator
set-up on the device, all_gather_into_tensor of the fixed (s, n_iter) records straight from the C library's device
result arrays, the two gathers of the variable-length CIGAR payload, barrier) executes once on the hardware; the
partition logic for world_size 2 is covered on CPU by tests/test_shard_gloo.py.  Runs in a child process (its own
process group, a hard timeout)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
from miniwfa_amd.shard import gather_records, gather_cigars, deal_pairs
from oracle.pyoracle import Oracle, make_opt

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
try:
    n = 24
    pairs = [synth_pair(98000 + i, 3000 if i %% 7 == 3 else 200 + 31 * (i %% 5), 0.06) for i in range(n)]
    deal = deal_pairs([len(t) + len(q) for t, q in pairs], dist.get_world_size())
    mine = [pairs[i] for i in deal[0]]
    pk = PackedBatch(mine)
    bufs = [torch.from_numpy(a.copy()).to(dev) for a in (pk.seqs, pk.t_off, pk.tl, pk.q_off, pk.ql)]
    eng = mw.Engine(0, torch.cuda.current_stream(dev).cuda_stream)
    b = eng.wrap(pk.n, bufs[0].data_ptr(), pk.total, bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(), bufs[4].data_ptr(), pk.tl, pk.ql, keep=tuple(bufs))
    b.align(mw.opt_init(flag=mw.MWF_F_CIGAR))
    s, it, nc = b.results()

    class DevPtr:
        def __init__(self, ptr, n, typestr):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    d_s = torch.as_tensor(DevPtr(b.dev_scores_ptr(), pk.n, "<i4"), device=dev)     # the library's device result arrays, zero-copy
    d_it = torch.as_tensor(DevPtr(b.dev_iters_ptr(), pk.n, "<i8"), device=dev)
    gs, git = gather_records(dist, d_s, d_it, n, device=dev, deal=deal)            # ONE RCCL all_gather_into_tensor on device memory
    assert gs.is_cuda and git.is_cuda
    cigs = gather_cigars(dist, [b.cigar(i, int(nc[i])) for i in range(pk.n)], n, device=dev, deal=deal)
    dist.barrier()
    torch.cuda.synchronize(dev)
    orc = Oracle()
    for i, (t, q) in enumerate(pairs):
        es, eit, ecig = orc.align(t, q, make_opt(flag=1))
        assert (int(gs[i]), int(git[i])) == (es, eit), i
        assert cigs[i].tolist() == ecig, i
    b.free()
    eng.close()
    print("RCCL-OK backend=%%s world=%%d" %% (dist.get_backend(), dist.get_world_size()))
finally:
    dist.destroy_process_group()
"""


def test_result_gather_runs_on_rccl_with_device_tensors():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "RCCL-OK backend=nccl world=1" in r.stdout
