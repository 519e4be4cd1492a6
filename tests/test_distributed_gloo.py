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
    """The collective of bench.py: all_gather of counts, then a max-padded gather to rank 0."""
    cnt = torch.tensor([len(ids)], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, cnt)
    mx = int(torch.stack(counts).max().item())
    send = torch.zeros(mx, dtype=torch.int32, device=device)
    send[:len(ids)] = torch.as_tensor(ids, dtype=torch.int32)
    bufs = [torch.empty(mx, dtype=torch.int32, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(send, bufs, dst=0)
    if rank != 0:
        return None
    return np.concatenate([bufs[r][:int(counts[r].item())].numpy() for r in range(world)])


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


@pytest.mark.parametrize("world", [2])
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
