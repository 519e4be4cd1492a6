#!/usr/bin/env python3
"""bench.py — input MB/s tokenized by the HIP Linear WordPiece path on N MI355X GPUs.

A step = one pass of the hot path (UTF-8 decode -> S build -> suffix array + LCP by prefix
doubling -> scanlines -> greedy walk -> id stream) over one shard per GPU, the shard already
resident in HBM and the ids left in HBM.  Workload = BASELINE.json configs[1]: a 100 MB
English-shaped shard with a 29k-line BERT-like vocabulary (synthetic, SURVEY.md §8d config 2;
enwiki and bert-base-cased vocab.txt are not available offline).  With N > 1 every rank
tokenizes its own shard (weak scaling, no collective on the data path) and the token ids are
gathered to rank 0 over RCCL at the end of every step, as the north star prescribes.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import wordpiece_amd as W  # noqa: E402
from wordpiece_amd import synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy rate
RADIX_BYTES_PER_ELEM = 24  # one radix pass reads and writes a 12-byte (key, index) record


_DevView = W.DeviceIds  # zero-copy torch view of a device buffer owned by the library


def _cpu_baseline(text, vocab, target_bytes):
    """The CPU port (oracle/, OpenMP; SA stage through the reference's own libsais when oracle/_ref
    was built) timed on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cut = min(len(text), target_bytes)
    while cut < len(text) and text[cut] not in b" \n":
        cut += 1
    sample = text[:cut]
    used_ref_sa = O.use_libsais(True)
    cores = os.cpu_count() or 1
    try:
        ov = O.Vocab(vocab)
        t0 = time.time()
        ids = ov.encode(sample, threads=cores)
        dt = time.time() - t0
    finally:
        O.use_libsais(False)
    return {"value": round(len(sample) / 1e6 / dt, 3), "unit": "MB/s", "cores": cores, "kind": "port",
            "sample": "first %.1f MB of the rank-0 shard, %d ids, %.1f s; oracle/wp_oracle.c with OpenMP, SA stage via %s"
                      % (len(sample) / 1e6, len(ids), dt,
                         "the reference's libsais built from source (oracle/_ref)" if used_ref_sa
                         else "the oracle's own prefix-doubling sorter")}, ids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mb", type=float, default=100.0, help="shard size per GPU in MB (1 MB = 1e6 bytes)")
    ap.add_argument("--vocab-size", type=int, default=29000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--cpu-sample-mb", type=float, default=32.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--text-file", default=None, help="optional local corpus instead of the synthetic shard")
    ap.add_argument("--vocab-file", default=None)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # WP_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks
    # share devices, collectives go through host memory); the driver's runs use nccl (= RCCL).
    backend = os.environ.get("WP_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where collective tensors live
    dist = None
    # WP_BENCH_FORCE_GATHER=1: run the collective path with a single rank too (checks the RCCL calls on a
    # one-GPU box; launch through torch.distributed.run --nproc-per-node 1)
    distributed = world > 1 or os.environ.get("WP_BENCH_FORCE_GATHER") == "1"
    if distributed:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- workload: one shard per rank ----
    nbytes_target = int(args.mb * 1e6)
    if args.text_file and args.vocab_file:
        with open(args.vocab_file, "rb") as f:
            vocab = f.read().split(b"\n")
            if vocab and vocab[-1] == b"":
                vocab.pop()
        with open(args.text_file, "rb") as f:
            data = f.read()
        s, e = W.shard_bounds(data, world)[rank]
        text = data[s:e][:nbytes_target] if nbytes_target else data[s:e]
        workload = "local files %s / %s" % (args.text_file, args.vocab_file)
    else:
        # one lexicon and one (replicated) vocabulary for all ranks; rank r > 0 draws its own word sequence
        text, vocab = synth.english_corpus(nbytes_target, seed=args.seed, vocab_size=args.vocab_size,
                                           text_seed=rank if rank > 0 else None)
        workload = ("configs[1]: %.0f MB English-shaped synthetic shard per GPU (SURVEY 8d config 2), "
                    "%d-line BERT-like vocab" % (args.mb, len(vocab)))
    nbytes = len(text)

    vocab_h = W.Vocab(vocab, device=dev_index)
    vocab_h.set_option(W.WP_OPT_STAGE_TIMING, 1)
    pad = (-nbytes) % 16 + 16
    d_text = torch.zeros(nbytes + pad, dtype=torch.uint8, device=dev)
    d_text[:nbytes] = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()

    # The only collective: token ids -> rank 0 (all_gather of the counts, gather of fixed-capacity
    # buffers).  Buffers are sized once, on the first (warm-up) step; after that the loop never reads a
    # device value on the host, so the gather of step i runs on RCCL's stream while the kernels of
    # step i+1 run on the encoder's streams.
    gather = None
    if distributed:
        from wordpiece_amd.gather import IdGather
        gather = IdGather(dist, rank, world, cdev)

    def step():
        d_ids, n_ids = vocab_h.encode_device(d_text.data_ptr(), nbytes)
        if distributed:
            view = (torch.as_tensor(_DevView(d_ids, n_ids), device=dev) if n_ids
                    else torch.zeros(0, dtype=torch.int32, device=dev))
            # the ids must be out of the handle's buffer before the next encode overwrites it
            gather.step(view, n_ids, before_collective=torch.cuda.current_stream().synchronize)
        return n_ids

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    stage_ms = {}
    radix_ms = radix_elems = radix_launches = 0
    t0 = time.perf_counter()
    n_ids = 0
    for _ in range(args.steps):
        n_ids = step()
        st = vocab_h.stats()
        radix_ms += st["ms_radix_scatter"]
        radix_elems += st["radix_pass_elems"]
        radix_launches += st["radix_passes"]
        for k in ("ms_total", "ms_decode", "ms_sa", "ms_lcp", "ms_scan", "ms_walk"):
            stage_ms[k] = stage_ms.get(k, 0.0) + st[k]
    fence()
    dt = time.perf_counter() - t0
    st = vocab_h.stats()

    t = torch.tensor([dt, float(nbytes)], dtype=torch.float64, device=cdev)
    if distributed:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, total_bytes = float(tmax[0].item()), float(tsum[1].item())
    else:
        dt_max, total_bytes = dt, float(nbytes)

    if rank == 0:
        ms_per_step = dt_max / args.steps * 1e3
        value = total_bytes / 1e6 / (dt_max / args.steps)
        # dominant kernel: the radix scatter pass; algorithmic bytes = 24 B per element moved
        avg_launch_ms = radix_ms / max(radix_launches, 1)
        # (the first pass of the round-0 sort makes the index column up instead of reading it: -4 B per symbol)
        avg_launch_bytes = (RADIX_BYTES_PER_ELEM * radix_elems - 4 * st["n_total"] * args.steps) / max(radix_launches, 1)
        achieved = avg_launch_bytes / 1e9 / (avg_launch_ms / 1e3) if avg_launch_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "radix_scatter_traffic.json")
        if os.path.exists(tfile) and abs(nbytes - 100_000_000) < 1_000_000:
            # PMC bytes per launch, collected separately (rocprofv3 --pmc) on this very workload
            with open(tfile) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "input MB/s tokenized (bert-base-cased vocab)", "value": round(value, 2), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": workload, "bytes_per_gpu": nbytes, "vocab_lines": len(vocab),
                       "symbols_n": st["n_total"], "ids_per_step_rank0": int(n_ids), "rounds": st["rounds"],
                       "sorted_depth": st["sorted_depth"], "symbol_bits": st["symbol_bits"],
                       "symbols_per_key": st["symbols_per_key"], "active_per_round": st["active_per_round"],
                       "radix_launches_per_step": radix_launches // max(args.steps, 1),
                       "id_gather": ("%s gather to rank 0" % ("rccl" if backend == "nccl" else backend)) if distributed
                       else "none (single GPU)"},
            "roofline": {"bound": "hbm", "kernel": "radix_scatter_kernel<uint64, 24> (full-size tiles; the round-0 suffix sort)", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "avg_launch_ms": round(avg_launch_ms, 4),
                         "algorithmic_bytes_per_launch": int(avg_launch_bytes)},
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage_ms.items()},
        }
        # the whole SA/LCP stage against the HBM peak (DESIGN.md section 4 lists the per-kernel bytes):
        # round-0 sort (32 B per element and pass: 8 histogram + 24 scatter, - 4 for the identity pass),
        # round-0 split 35, rank store 32, rounds >= 1 per active entry: LDS sort 28 + split 50 + rank store 32
        n_sym, act = st["n_total"], st["active_per_round"]
        sa_bytes = 32 * (radix_elems / max(args.steps, 1)) - 4 * n_sym + (35 + 32) * n_sym + 110 * sum(act[1:])
        sa_ms = stage_ms.get("ms_sa", 0.0) / max(args.steps, 1)
        if sa_ms > 0:
            out["sa_lcp_stage"] = {"algorithmic_bytes": int(sa_bytes), "ms": round(sa_ms, 3),
                                   "achieved": round(sa_bytes / 1e9 / (sa_ms / 1e3), 1), "unit": "GB/s",
                                   "frac": round(sa_bytes / 1e9 / (sa_ms / 1e3) / HBM_PEAK_GBPS, 4)}
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_ids = _cpu_baseline(text, vocab, int(args.cpu_sample_mb * 1e6))
            out["cpu_baseline"] = cb
            # the sample is a whitespace-cut prefix of the shard: its ids are a prefix of the GPU's
            d_ids, n = vocab_h.encode_device(d_text.data_ptr(), nbytes)
            gpu_ids = torch.as_tensor(_DevView(d_ids, n), device=dev)[:len(cpu_ids)].cpu().numpy()
            out["config"]["ids_match_cpu_port_on_sample"] = bool(np.array_equal(gpu_ids, cpu_ids))
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
