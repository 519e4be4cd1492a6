"""Token-id gather of the multi-GPU path: the only collective.

Every rank tokenizes its own shard; the id streams are collected on rank 0 in shard order with exact
sizes: an all_gather of the counts, then one point-to-point receive per peer straight into its place
in rank 0's buffer (RCCL send/recv over the peer's own xGMI link; no max-padded gather, so rank 0
holds sum(counts) ids and nothing else).  The buffer grows on demand and is reused across steps.
Used by bench.py (backend nccl = RCCL, or gloo for rehearsals) and tests/test_distributed_gloo.py.
(The single-process form of the same thing is wp_linear_encode_multi behind the C ABI: one host thread
and context per GPU, ids downloaded into one host buffer.)
"""
import torch


class IdGather:
    def __init__(self, dist, rank, world, device):
        self.dist, self.rank, self.world, self.device = dist, rank, world, device
        self.cnt = torch.zeros(1, dtype=torch.int64, device=device)
        self.counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        self.buf = None     # rank 0: all ids in shard order
        self.send = None    # other ranks: staging copy of the own ids (the encoder reuses its buffer)
        self.host_counts = [0] * world
        self.primed = False  # host_counts hold the counts of a step

    def _room(self, t, n):
        if t is None or t.numel() < n:
            t = torch.empty(int(n * 1.05) + 1024, dtype=torch.int32, device=self.device)
        return t

    def step(self, ids, n_ids, before_collective=None, counts_known=False):
        """ids: int32 tensor (any device) holding n_ids ids.  before_collective(): called after the ids
        have been copied out of `ids` (bench.py synchronises there, because the encoder reuses the buffer).
        counts_known: EVERY rank promises that its count is the one of the previous step (the same shard encoded
        again): the count exchange and its read-back — a host sync per step — are skipped and the receives are
        posted with the sizes already known.  A rank whose count did change raises instead of posting a receive
        of the wrong size (the decision has to be the same on all ranks, so there is no silent fallback)."""
        dist = self.dist
        if counts_known and self.primed:
            if n_ids != self.host_counts[self.rank]:
                raise RuntimeError("IdGather: counts_known, but this rank's count changed (%d -> %d)"
                                   % (self.host_counts[self.rank], n_ids))
        else:
            self.cnt.fill_(n_ids)
            dist.all_gather(self.counts, self.cnt)
            self.host_counts = [int(c) for c in torch.cat(self.counts).tolist()]  # one read-back for all counts
            self.primed = True
        if self.rank == 0:
            total = sum(self.host_counts)
            self.buf = self._room(self.buf, total)
            if n_ids:
                self.buf[:n_ids].copy_(ids[:n_ids])
        else:
            self.send = self._room(self.send, n_ids)
            if n_ids:
                self.send[:n_ids].copy_(ids[:n_ids])
        if before_collective:
            before_collective()
        if self.world == 1:
            return
        ops = []
        if self.rank == 0:
            off = self.host_counts[0]
            for r in range(1, self.world):
                if self.host_counts[r]:
                    ops.append(dist.P2POp(dist.irecv, self.buf[off:off + self.host_counts[r]], r))
                off += self.host_counts[r]
        elif n_ids:
            ops.append(dist.P2POp(dist.isend, self.send[:n_ids], 0))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def result(self):
        """rank 0: the ids of all shards in shard order (numpy int32); other ranks: None."""
        if self.rank != 0:
            return None
        return self.buf[:sum(self.host_counts)].cpu().numpy()
