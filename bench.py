#!/usr/bin/env python3
"""bench.py — input MB/s tokenized by the HIP Linear WordPiece path on N MI355X GPUs.

A step = one pass of the hot path (UTF-8 decode -> S build -> suffix array + LCP by prefix
doubling -> scanlines -> greedy walk -> id stream) over one shard per GPU, the shard already
resident in HBM and the ids left in HBM (`value`).  Default workload = BASELINE.json configs[1]:
a 100 MB English-shaped shard with a 29k-line BERT-like vocabulary (synthetic, SURVEY.md §8d
config 2; enwiki and bert-base-cased vocab.txt are not available offline).  `--config 3|4|5`
selects the other single-GPU-sized configurations (1 GB mixed scripts, one 1.25 GB shard of the
10 GB run per GPU, 1 GB deep-prefix stress).  With N > 1 every rank tokenizes its own shard (weak
scaling, no collective on the data path) and the token ids are gathered to rank 0 over RCCL at the end
of every step, as the north star prescribes.

Also reported on rank 0 at N = 1: `host_to_host` (SURVEY §8d's metric proper: host-resident text to
host-resident ids through wp_linear_encode, upload and download included), the sibling `fast` path's
device rate, the roofline of the dominant kernel, the SA/LCP stage against the HBM peak and the CPU
baseline.  Prints ONE JSON line on rank 0; exits non-zero if the ids differ from the CPU port's.
"""
import argparse
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

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBPS = 6300.0  # what plain read / copy kernels reach on this GPU (guide: ~6.3; profiles/r01_hbm_probe.txt: 6.25 / 5.73)
# one radix pass reads and writes a (key, index) record: 12 bytes with a 64-bit key (round 1 of this
# build), 8 bytes with the 32-bit round-0 keys of this round (wp_stats.key_bits); radix_bytes() below
# SA/LCP stage, algorithmic bytes per symbol besides the radix passes:
SPLIT_BYTES = 12       # round 0 after the sort (round0_rank_kernel): keys 4 read, rank 4 + LCP 4 written (no LCP array
                       # in the text-only layout, where nothing reads it: 8)
RANK_STORE_BYTES = 12  # window store: (destination, rank) 8 read, rank 4 written; the one or two partition passes in
                       # front of it are full-size launches of the radix scatter (16 B per element, counted there)
ROUND_BYTES = 110      # rounds >= 1, per list entry: LDS sort 28 + split 50 + rank store 32

# In-container calibration of the CPU port against the compiled reference (filled in from
# DESIGN.md section 4): same config-2 synthetic, 8 vCPUs of the build container.
CPU_CALIBRATION = ("build container (8 vCPU Xeon 2.1 GHz), config-2 synthetic 100 MB, idle: CPU port 43.4 s on 8 threads / 42.1 s on 1 "
                   "(24 MB: 2.7 s / 7.0 s); compiled reference per SURVEY 8(c) probe: 17.4 s on 8 vCPU / 62.7 s on 1 pool thread; "
                   "reference's published laptops: 9.4-18.9 MB/s")

_DevView = W.DeviceIds  # zero-copy torch view of a device buffer owned by the library

CONFIGS = {
    2: ("english", 100.0, 29000, "configs[1]: %.0f MB English-shaped synthetic shard per GPU (SURVEY 8d config 2)"),
    3: ("multilingual", 1000.0, 120000, "configs[2]: %.0f MB mixed en/ru/ja/zh synthetic text per GPU, 120k-line vocab (SURVEY 8d config 3)"),
    4: ("english", 1250.0, 29000, "configs[3]: %.0f MB English-shaped shard per GPU = one of the 8 shards of the 10 GB run (SURVEY 8d config 4)"),
    5: ("deep", 1000.0, 0, "configs[4]: %.0f MB of 512-char words, every stem prefix a token (SURVEY 8d config 5)"),
}


def _host_cpu():
    """CPU model string, physical cores and logical CPUs of the box (Linux /proc/cpuinfo), and the CPUs this
    process may run on."""
    model, phys, logical = "unknown", set(), 0
    try:
        with open("/proc/cpuinfo") as f:
            pid = None
            for ln in f:
                k, _, v = ln.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name" and model == "unknown":
                    model = v
                elif k == "processor":
                    logical += 1
                elif k == "physical id":
                    pid = v
                elif k == "core id":
                    phys.add((pid, v))
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return model, (len(phys) or logical or usable), (logical or usable), usable


def _cpu_baseline(text, vocab, target_bytes):
    """The CPU port (oracle/, OpenMP; SA stage through the reference's own libsais when oracle/_ref
    was built) timed on bounded samples of the same workload, as SURVEY 8(d) asks: all physical cores the
    process may use (capped at 64: a 64 MB sample gives more threads nothing to do) and one thread, CPU model
    and core counts stated."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    def sample_of(nbytes):
        cut = min(len(text), int(nbytes))
        while cut < len(text) and text[cut] not in b" \n":
            cut += 1
        return text[:cut]

    model, phys, logical, usable = _host_cpu()
    threads = max(1, min(phys, usable, 64))
    sample = sample_of(target_bytes)
    one = sample_of(max(target_bytes / 8, 2e6))
    used_ref_sa = O.use_libsais(True)
    try:
        ov = O.Vocab(vocab)
        t0 = time.time()
        ids = ov.encode(sample, threads=threads)
        dt = time.time() - t0
        t0 = time.time()
        ids1 = ov.encode(one, threads=1)
        dt1 = time.time() - t0
    finally:
        O.use_libsais(False)
    sa = "the reference's libsais built from source (oracle/_ref)" if used_ref_sa else "the oracle's own prefix-doubling sorter"
    return {"value": round(len(sample) / 1e6 / dt, 3), "unit": "MB/s", "cores": threads, "kind": "port",
            "sample": "first %.1f MB of the rank-0 shard, %d ids, %.1f s on %d threads; oracle/wp_oracle.c with OpenMP, SA stage via %s"
                      % (len(sample) / 1e6, len(ids), dt, threads, sa),
            "single_thread": {"value": round(len(one) / 1e6 / dt1, 3), "unit": "MB/s", "cores": 1,
                              "sample": "first %.1f MB, %d ids, %.1f s" % (len(one) / 1e6, len(ids1), dt1)},
            "cpu_model": model, "physical_cores": phys, "logical_cpus": logical, "usable_cpus": usable,
            "calibration": CPU_CALIBRATION}, ids


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as children
    (`python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`, the driver's own form),
    relay rank 0's single JSON line and fail unless all N ranks took part.  The reference's precedent for fanning
    out inside one call is linear.cpp:283-299.  Runs before this process makes any GPU call and never exec()s."""
    import socket
    import subprocess

    backend = os.environ.get("WP_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if backend == "nccl" and have < n:
        print("bench.py: --gpus %d but only %d GPU(s) visible (WP_BENCH_BACKEND=gloo rehearses with shared devices)"
              % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)  # anything else a child wrote to stdout
    if r.returncode != 0 or line is None:
        print("bench.py: the %d-rank run failed (exit code %d)" % (n, r.returncode), file=sys.stderr)
        return r.returncode or 1
    got = json.loads(line).get("n_gpus")
    if got != n:
        print("bench.py: %d rank(s) reported, %d asked for" % (got, n), file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="SURVEY 8d configuration (2 = the metric's)")
    ap.add_argument("--mb", type=float, default=None, help="shard size per GPU in MB (1 MB = 1e6 bytes); default: the configuration's")
    ap.add_argument("--vocab-size", type=int, default=None)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--cpu-sample-mb", type=float, default=64.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-to-host and fast-path measurements")
    ap.add_argument("--text-file", default=None, help="optional local corpus instead of the synthetic shard")
    ap.add_argument("--vocab-file", default=None)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it has not touched the GPU and
        # never will), N fresh ranks do the work — never a silent single-GPU run under an N-GPU label
        raise SystemExit(_self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if torch.cuda.device_count() == 0:  # (counting devices does not initialise the GPU: the corpus workers fork below)
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")

    # ---- workload: one shard per rank (generated before the GPU is touched: worker processes) ----
    kind, mb_default, vs_default, wl_fmt = CONFIGS[args.config]
    mb = args.mb if args.mb is not None else mb_default
    vocab_size = args.vocab_size if args.vocab_size is not None else vs_default
    nbytes_target = int(mb * 1e6)
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
        text, vocab = synth.parallel_corpus(kind, nbytes_target, args.seed, vocab_size, rank)
        workload = (wl_fmt % mb) + ", %d-line %s vocab" % (len(vocab), "BERT-like" if kind != "deep" else "prefix")
    nbytes = len(text)

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
        import contextlib
        import torch.distributed as dist

        @contextlib.contextmanager
        def stdout_to_stderr():
            """RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for
            the ONE JSON line (file-descriptor level: the banner comes from the C library)."""
            sys.stdout.flush()
            saved = os.dup(1)
            try:
                os.dup2(2, 1)
                yield
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)

        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            dist.barrier()  # (brings the communicator up here, banner included)

    vocab_h = W.Vocab(vocab, device=dev_index)
    vocab_h.set_option(W.WP_OPT_STAGE_TIMING, 1)
    vocab_h.reserve(nbytes)
    pad = (-nbytes) % 16 + 16
    d_text = torch.zeros(nbytes + pad, dtype=torch.uint8, device=dev)
    d_text[:nbytes] = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()

    # The only collective: token ids -> rank 0 (exact counts by all_gather, then one receive per peer into
    # its place in a buffer sized once).  After the first step the loop never reads a device value on the
    # host, so the gather of step i runs on RCCL's stream while the kernels of step i+1 run on the encoder's.
    gather = None
    if distributed:
        from wordpiece_amd.gather import IdGather
        gather = IdGather(dist, rank, world, cdev)

    steps_done = [0]

    def step():
        d_ids, n_ids = vocab_h.encode_device(d_text.data_ptr(), nbytes)
        if distributed:
            view = (torch.as_tensor(_DevView(d_ids, n_ids), device=dev) if n_ids
                    else torch.zeros(0, dtype=torch.int32, device=dev))
            # the ids must be out of the handle's buffer before the next encode overwrites it; every step encodes
            # the same shard, so after the first one the counts are known and nothing is read back from the device
            gather.step(view, n_ids, before_collective=torch.cuda.current_stream().synchronize,
                        counts_known=steps_done[0] > 0)
        steps_done[0] += 1
        return n_ids

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    stage_ms = {}
    radix_ms = radix_elems = radix_launches = digit_bytes = radix_bytes = 0
    t0 = time.perf_counter()
    n_ids = 0
    for _ in range(args.steps):
        n_ids = step()
        st = vocab_h.stats()
        radix_ms += st["ms_radix_scatter"]
        radix_elems += st["radix_pass_elems"]
        radix_launches += st["radix_passes"]
        digit_bytes += st["radix_digit_bytes"]
        radix_bytes += st["radix_pass_bytes"]
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

    ok = True
    if rank == 0:
        steps = max(args.steps, 1)
        key_bytes = 4 if 0 < st["key_bits"] <= 32 else 8
        RADIX_BYTES_PER_ELEM = 2 * (key_bytes + 4)
        scatter_name = "radix_scatter_kernel<%s, stable|first-pass> (full-size tiles: the %d passes of the round-0 suffix sort over %d-bit keys%s; %d-byte records)" % (
            "uint32, 20" if key_bytes == 4 else "uint64, 24", (st["key_bits"] + 7) // 8, st["key_bits"],
            " and the two destination-partition passes of the round-0 rank store (the first one makes its value column - the slots - up)"
            if key_bytes == 4 else "", key_bytes + 4)
        ms_per_step = dt_max / args.steps * 1e3
        value = total_bytes / 1e6 / (dt_max / args.steps)
        # dominant kernel: the radix scatter pass.  Algorithmic bytes per launch (counted by the library per launch,
        # wp_stats.radix_pass_bytes) = the (key, index) record read and written per element moved (16 B with 32-bit
        # keys), - 4 B per element where the pass makes the index column up instead of reading it (the first pass of the
        # suffix sort and of the rank store), + 1 B per element for the digit byte it leaves for the next pass's histogram
        avg_launch_ms = radix_ms / max(radix_launches, 1)
        avg_launch_bytes = radix_bytes / max(radix_launches, 1)
        achieved = avg_launch_bytes / 1e9 / (avg_launch_ms / 1e3) if avg_launch_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "radix_scatter_traffic.json")
        if os.path.exists(tfile) and args.config == 2 and abs(nbytes - 100_000_000) < 1_000_000:
            # PMC bytes per launch, collected separately (rocprofv3 --pmc) on this very workload
            with open(tfile) as f:
                tj = json.load(f)
            if abs(tj.get("algorithmic_bytes_per_launch", 0) - avg_launch_bytes) < 0.02 * avg_launch_bytes:
                traffic = tj.get("hbm_bytes_per_launch")
        out = {
            "metric": "input MB/s tokenized (bert-base-cased vocab; offline stand-in: synthetic BERT-like vocab)",
            "value": round(value, 2), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": workload, "value_is": "device-resident: text already in HBM, ids left in HBM (host_to_host: the same through host buffers)",
                       "bytes_per_gpu": nbytes, "vocab_lines": len(vocab),
                       "symbols_n": st["n_total"], "n_text": st["n_text"], "vocab_in_s": st["vocab_in_s"],
                       "ids_per_step_rank0": int(n_ids), "rounds": st["rounds"],
                       "sorted_depth": st["sorted_depth"], "symbol_bits": st["symbol_bits"],
                       "symbols_per_key": st["symbols_per_key"], "key_bits": st["key_bits"], "active_per_round": st["active_per_round"],
                       "needed_after_round0": st["needed_after_round0"],
                       "radix_launches_per_step": radix_launches // steps, "staged_emit": st["staged_emit"],
                       "trie_refine": st["trie_refine"], "hist_in_keys": st["hist_in_keys"], "list_retries": st["list_retries"],
                       "arena_bytes": st["arena_bytes"], "arena_bytes_per_symbol": round(st["arena_bytes"] / max(st["n_total"], 1), 1),
                       "id_gather": ("%s: exact-size receives on rank 0" % ("rccl" if backend == "nccl" else backend)) if distributed
                       else "none (single GPU)"},
            "roofline": {"bound": "hbm", "kernel": scatter_name, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "achievable": HBM_ACHIEVABLE_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBPS, 4),
                         "traffic": traffic, "avg_launch_ms": round(avg_launch_ms, 4),
                         "algorithmic_bytes_per_launch": int(avg_launch_bytes)},
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage_ms.items()},
        }
        # the whole SA/LCP stage against the HBM peak, SURVEY 8(d) style: algorithmic bytes only (no histogram
        # re-read of the keys: the histograms read the digit bytes)
        n_sym, act = st["n_total"], st["active_per_round"]
        # counted scatter launches (records + the digit bytes they write) + the digit bytes the histograms read (every
        # byte written once is read once; the first pass's bytes come from the key builder - written + read - unless it
        # took the first histogram itself), or the keys the histograms re-read where there are no digit bytes
        pass_bytes = radix_bytes / steps
        dig_read = (digit_bytes / steps + (0 if st["hist_in_keys"] else 2 * n_sym)) if digit_bytes else key_bytes * radix_elems / steps
        # no rank kernel of its own where a suffix's slot is its rank (trie_refine: nothing asks for group heads)
        split = 0 if st["trie_refine"] else SPLIT_BYTES - (4 if st["vocab_in_s"] == 0 else 0)
        sa_bytes = pass_bytes + dig_read + (split + RANK_STORE_BYTES) * n_sym + ROUND_BYTES * sum(act[1:])
        dig = dig_read + (digit_bytes / steps if digit_bytes else 0)
        sa_ms = stage_ms.get("ms_sa", 0.0) / steps
        if sa_ms > 0:
            out["sa_lcp_stage"] = {"algorithmic_bytes": int(sa_bytes), "ms": round(sa_ms, 3),
                                   "achieved": round(sa_bytes / 1e9 / (sa_ms / 1e3), 1), "unit": "GB/s",
                                   "frac": round(sa_bytes / 1e9 / (sa_ms / 1e3) / HBM_PEAK_GBPS, 4),
                                   "per_symbol": {"radix_pass": RADIX_BYTES_PER_ELEM, "digit_bytes_total": round(dig / n_sym, 2),
                                                  "split": split, "rank_store_window": RANK_STORE_BYTES, "round_entry": ROUND_BYTES}}
        if world == 1 and not args.no_extras:
            # SURVEY 8(d)'s metric proper: host-resident text -> host-resident ids (pageable source, ids into a pinned
            # block of the library's pool), wall clock around wp_linear_encode
            vocab_h.encode(text[:1 << 20])
            reps = max(2, min(args.steps, 5))
            h2d = d2h = 0.0
            t1 = time.perf_counter()
            for _ in range(reps):
                ids_h = vocab_h.encode(text)
                hs = vocab_h.stats()
                h2d += hs["ms_h2d"]
                d2h += hs["ms_d2h"]
                del ids_h
            hw = (time.perf_counter() - t1) / reps
            out["host_to_host"] = {"ms_per_step": round(hw * 1e3, 3), "MB_per_s": round(nbytes / 1e6 / hw, 1),
                                   "h2d_ms": round(h2d / reps, 3), "d2h_ms": round(d2h / reps, 3),
                                   "note": "h2d_ms is host time of the (synchronous, pageable-source) upload; d2h into pinned memory"}
            # the same through the shard pipeline (wp_linear_encode_batch): a sequence of shards of one corpus, uploads and
            # id downloads on copy streams beside the neighbouring shard's kernels
            k = 8
            got = []
            vocab_h.encode_stream([text] * 2, lambda i, ids: None)  # (warm: second text buffer, staging buffers, pinned blocks)
            t1 = time.perf_counter()
            vocab_h.encode_stream([text] * k, lambda i, ids: got.append((i, len(ids), int(ids[0]) if len(ids) else -1)))
            bw = (time.perf_counter() - t1) / k
            bs = vocab_h.stats()
            same_ids = [g[0] for g in got] == list(range(k)) and all(g[1] == int(n_ids) for g in got)
            out["host_to_host_stream"] = {"ms_per_shard": round(bw * 1e3, 3), "MB_per_s": round(nbytes / 1e6 / bw, 1), "shards": k,
                                          "ids_per_shard_as_single_call": same_ids,
                                          "upload_ms_per_shard": round(bs["ms_h2d"] / k, 3), "device_ms_per_shard": round(bs["ms_total"] / k, 3),
                                          "note": "wp_linear_encode_stream: %d shards of this size back to back, pageable sources, ids delivered in two pinned blocks taking turns" % k}
            # the sibling fast path (word_piece::fast), device resident
            vocab_h.fast_encode_device(d_text.data_ptr(), nbytes)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                f_ptr, f_n = vocab_h.fast_encode_device(d_text.data_ptr(), nbytes)
            fw = (time.perf_counter() - t1) / reps
            fast_ids = torch.as_tensor(_DevView(f_ptr, f_n), device=dev).clone() if f_n else None
            l_ptr, l_n = vocab_h.encode_device(d_text.data_ptr(), nbytes)
            same = bool(f_n == l_n and (f_n == 0 or torch.equal(fast_ids, torch.as_tensor(_DevView(l_ptr, l_n), device=dev))))
            out["fast_path"] = {"ms_per_step": round(fw * 1e3, 3), "MB_per_s": round(nbytes / 1e6 / fw, 1),
                                "ids_equal_linear_on_device": same}
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_ids = _cpu_baseline(text, vocab, int(args.cpu_sample_mb * 1e6))
            out["cpu_baseline"] = cb
            # the sample is a whitespace-cut prefix of the shard: its ids are a prefix of the GPU's
            d_ids, n = vocab_h.encode_device(d_text.data_ptr(), nbytes)
            gpu_ids = torch.as_tensor(_DevView(d_ids, n), device=dev)[:len(cpu_ids)].cpu().numpy()
            ok = bool(np.array_equal(gpu_ids, cpu_ids))
            out["config"]["ids_match_cpu_port_on_sample"] = ok
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bench.py: the HIP path's ids differ from the CPU port's on the sample")


if __name__ == "__main__":
    main()
