"""world_size-2 CPU rehearsal (gloo) of the multi-GPU path: whitespace sharding + the id gather
bench.py performs over RCCL.  The per-shard encoder is the CPU oracle here (checker only) — the
point of the test is that sharded ids, gathered in shard order, equal the unsharded ids."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def gather_ids(ids, rank, world, device="cpu"):
    """The collective of bench.py (wordpiece_amd/gather.py), two steps to cover the buffer reuse."""
    from wordpiece_amd.gather import IdGather
    g = IdGather(dist, rank, world, torch.device(device))
    t = torch.as_tensor(np.ascontiguousarray(ids), dtype=torch.int32)
    g.step(t, len(ids))
    g.step(t, len(ids))
    return g.result()


def _worker(rank, world, port, text, vocab, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import wordpiece_amd as W
    s, e = W.shard_bounds(text, world)[rank]
    ids = O.Vocab(vocab).encode(text[s:e])
    allids = gather_ids(ids, rank, world)
    if rank == 0:
        np.save(out_path, allids)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gather_equals_unsharded(tmp_path, world):
    import oracle_lib as O
    from wordpiece_amd import synth
    text, vocab = synth.english_corpus(400_000, seed=4, vocab_size=3000)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "ids.npy")
    mp.spawn(_worker, args=(world, port, text, vocab, out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), O.Vocab(vocab).encode(text))
