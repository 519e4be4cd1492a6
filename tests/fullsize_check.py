"""Full-size runs of BASELINE.json's configs on one MI355X with size-independent checks.

usage (on the GPU box):  python tests/fullsize_check.py <english|multilingual|deep> <bytes> [out.json]
(not collected by pytest: minutes of corpus generation; results are kept under profiles/)

The corpus is generated in ~100 MB chunks by worker processes (before the GPU is touched), the
vocabulary comes from chunk 0.  Checks:
  1. shard property: ids(whole text) == concatenation of ids(shard) over ~100 MB
     shards cut at whitespace — the 1 GB code path against the 100 MB code path;
  2. the first 16 MB (cut at whitespace) against the CPU oracle (test infrastructure, tests/oracle_lib.py).
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from wordpiece_amd import synth  # noqa: E402


def gen(args):
    kind, nbytes, k = args
    if kind == "english":
        text, vocab = synth.english_corpus(nbytes, seed=100 + k)
    elif kind == "multilingual":
        text, vocab = synth.multilingual_corpus(nbytes, seed=200 + 7 * k, vocab_size=120000)
    else:
        text, vocab = synth.deep_prefix_corpus(nbytes, seed=300, words_seed=k)
    if not text.endswith((b" ", b"\n")):
        text += b"\n"
    return text, (vocab if k == 0 else None)


def main():
    kind, total = sys.argv[1], int(float(sys.argv[2]))
    out = sys.argv[3] if len(sys.argv) > 3 else None
    nchunks = max(1, total // 100_000_000)
    t0 = time.time()
    with mp.Pool(min(nchunks, 12)) as pool:
        parts = pool.map(gen, [(kind, total // nchunks, k) for k in range(nchunks)])
    vocab = parts[0][1]
    text = b"".join(p[0] for p in parts)
    del parts
    print("generated %d bytes, vocab %d lines in %.0f s" % (len(text), len(vocab), time.time() - t0), flush=True)

    import wordpiece_amd as W  # the GPU is first touched here, after the workers are gone
    gv = W.Vocab(vocab)
    gv.encode(text[:1_000_000])  # warm-up
    t0 = time.time()
    ids = gv.encode(text)
    wall = time.time() - t0
    st = gv.stats()
    print("whole: %d ids, %.1f ms wall (host buffers), n=%d rounds=%d" % (len(ids), wall * 1e3, st["n_total"], st["rounds"]),
          flush=True)
    gv.set_option(W.WP_OPT_STAGE_TIMING, 1)
    gv.encode(text)
    st = gv.stats()
    bounds = W.shard_bounds(text, nchunks)
    pos, ok = 0, True
    for a, b in bounds:
        s_ids = gv.encode(text[a:b])
        if not np.array_equal(s_ids, ids[pos:pos + len(s_ids)]):
            ok = False
            print("MISMATCH in shard", a, b, flush=True)
            break
        pos += len(s_ids)
    ok = ok and pos == len(ids)
    print("shard property:", "ok" if ok else "FAILED", flush=True)

    import oracle_lib as O
    cut = min(len(text), 16_000_000)
    while cut < len(text) and text[cut] not in (9, 10, 11, 12, 13, 32):
        cut += 1
    O.use_libsais(True)
    t0 = time.time()
    o_ids = O.Vocab(vocab).encode(text[:cut], threads=os.cpu_count() or 8)
    print("oracle on %d bytes: %.1f s" % (cut, time.time() - t0), flush=True)
    g_ids = gv.encode(text[:cut])
    ok2 = np.array_equal(np.asarray(o_ids, dtype=np.int32), g_ids)
    print("oracle prefix check:", "ok" if ok2 else "FAILED", flush=True)
    res = {"config": kind, "bytes": len(text), "vocab_lines": len(vocab), "n_ids": int(len(ids)),
           "symbols_n": st["n_total"], "alphabet": st["alphabet"], "rounds": st["rounds"],
           "sorted_depth": st["sorted_depth"], "longest_token": st["longest_token"],
           "device_ms": st.get("ms_total"), "stage_ms": {k: st[k] for k in st if k.startswith("ms_")},
           "device_MB_per_s": (len(text) / 1e6) / (st["ms_total"] / 1e3) if st.get("ms_total") else None,
           "host_buffer_wall_ms": wall * 1e3, "shards_checked": nchunks, "shard_property_ok": bool(ok),
           "oracle_prefix_bytes": cut, "oracle_prefix_ok": bool(ok2)}
    print(json.dumps(res), flush=True)
    if out:
        with open(out, "w") as f:
            f.write(json.dumps(res) + "\n")
    sys.exit(0 if ok and ok2 else 1)


if __name__ == "__main__":
    main()
