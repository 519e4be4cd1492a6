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
    g.step(t, len(ids), counts_known=True)  # bench.py's steady state: no count exchange, no read-back
    if world > 1 and rank == world - 1 and len(ids) > 1:
        # a changed count under counts_known must raise, never post a transfer of the wrong size
        with pytest.raises(RuntimeError, match="count changed"):
            g.step(t[:-1], len(ids) - 1, counts_known=True)
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


def _run_bench(args, env_extra, timeout=600):
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True,
                          env=env, timeout=timeout)


def test_bench_gpus_n_never_runs_as_one_rank():
    """`python bench.py --gpus N` without a launcher starts the N ranks itself (the reference fans out
    inside the call too, linear.cpp:283-299) and fails loudly when it cannot — it never prints a line measured
    on one GPU.  Without a GPU (this test-suite's CPU leg) every form must exit non-zero and print no JSON."""
    if torch.cuda.device_count() > 0:
        pytest.skip("CPU leg: the launcher's success path is test_gpu_api.py::test_bench_self_launch_two_ranks")
    r = _run_bench(["--gpus", "2", "--mb", "1"], {})  # backend nccl: not enough GPUs
    assert r.returncode == 2 and "only 0 GPU(s) visible" in r.stderr and r.stdout.strip() == ""
    r = _run_bench(["--gpus", "2", "--mb", "1"], {"WP_BENCH_BACKEND": "gloo"})  # two ranks start, none finds a GPU
    assert r.returncode != 0 and "the 2-rank run failed" in r.stderr and r.stdout.strip() == ""
    assert "bench.py needs an MI355X" in r.stderr  # a rank came up and said so (the launcher stops the other one)
    r = _run_bench(["--gpus", "2", "--mb", "1"], {"WORLD_SIZE": "1", "RANK": "0"})  # a launcher with the wrong size
    assert r.returncode != 0 and "WORLD_SIZE 1 != --gpus 2" in r.stderr
