"""Token-id gather of the multi-GPU path: the only collective.

Every rank tokenizes its own shard; the id streams are collected on rank 0 in shard order.
`IdGather` sizes fixed-capacity buffers on the first step and afterwards never reads a device value
on the host, so over RCCL the gather of step i overlaps the kernels of step i+1.
Used by bench.py (backend nccl = RCCL, or gloo for rehearsals) and tests/test_distributed_gloo.py.
"""
import torch


class IdGather:
    def __init__(self, dist, rank, world, device):
        self.dist, self.rank, self.world, self.device = dist, rank, world, device
        self.cap = None

    def _size(self, n_ids):
        cnt = torch.tensor([n_ids], dtype=torch.int64, device=self.device)
        counts = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
        self.dist.all_gather(counts, cnt)
        self.cap = int(int(torch.stack(counts).max().item()) * 1.1) + 1024
        self.cnt, self.counts = cnt, counts
        self.send = torch.zeros(self.cap, dtype=torch.int32, device=self.device)
        self.recv = ([torch.empty(self.cap, dtype=torch.int32, device=self.device) for _ in range(self.world)]
                     if self.rank == 0 else None)

    def step(self, ids, n_ids, before_collective=None):
        """ids: int32 tensor (any device) holding n_ids ids.  before_collective(): called after the ids
        have been copied out of `ids` (bench.py synchronises there, because the encoder reuses the buffer)."""
        if self.cap is None:
            self._size(n_ids)
        if n_ids > self.cap:
            raise RuntimeError("id count %d exceeds the gather capacity %d" % (n_ids, self.cap))
        self.cnt.fill_(n_ids)
        if n_ids:
            self.send[:n_ids].copy_(ids[:n_ids] if ids.device == self.send.device else ids[:n_ids].to(self.send.device))
        if before_collective:
            before_collective()
        self.dist.all_gather(self.counts, self.cnt)
        self.dist.gather(self.send, self.recv, dst=0)

    def result(self):
        """rank 0: the ids of all shards in shard order (numpy int32); other ranks: None."""
        if self.rank != 0:
            return None
        import numpy as np
        return np.concatenate([self.recv[r][:int(self.counts[r].item())].cpu().numpy() for r in range(self.world)])
