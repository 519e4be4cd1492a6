"""Host-to-host rate of one 100 MB shard: one context against several contexts on the SAME GPU
(wp_linear_encode_multi with the device listed k times: k host threads, each uploads / encodes / downloads its
whitespace-cut piece, so that one piece's copies overlap another piece's kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from wordpiece_amd import synth
import wordpiece_amd as W

nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
text, vocab = synth.english_corpus(nbytes, seed=1234, vocab_size=29000)
gv = W.Vocab(vocab)
ref = gv.encode(text)
for k in (1, 2, 3, 4):
    devs = [0] * k
    ids = gv.encode_multi(text, devs)
    assert np.array_equal(ids, ref)
    del ids
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        ids = gv.encode_multi(text, devs)
        t.append(time.perf_counter() - t0)
        del ids
    print("%d context(s) on one GPU: host to host best %.2f ms, median %.2f ms = %.2f GB/s" % (
        k, min(t) * 1e3, sorted(t)[2] * 1e3, len(text) / sorted(t)[2] / 1e9), flush=True)
t = []
for _ in range(5):
    t0 = time.perf_counter()
    ids = gv.encode(text)
    t.append(time.perf_counter() - t0)
    del ids
print("wp_linear_encode: median %.2f ms = %.2f GB/s" % (sorted(t)[2] * 1e3, len(text) / sorted(t)[2] / 1e9))
