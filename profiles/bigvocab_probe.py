"""Large vocabularies: stage times with 30 k .. 1 M vocab lines on a 50 MB English-shaped text."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import numpy as np
from wordpiece_amd import synth
import wordpiece_amd as W
text, vocab = synth.english_corpus(50_000_000, seed=5, vocab_size=29000, lexicon_size=600000)
rng = np.random.default_rng(1)
base = set(vocab)
letters = "etaoinshrdlcumwfgypbvkjxqz"
extra = []
while len(extra) < 1_000_000:
    k = int(rng.integers(3, 10))
    w = "".join(letters[i] for i in rng.integers(0, 26, k))
    if rng.random() < 0.5:
        w = "##" + w
    if w not in base:
        base.add(w)
        extra.append(w)
for nv in (29000, 250_000, 1_000_000):
    v = vocab + extra[: nv - len(vocab)]
    t0 = time.time(); gv = W.Vocab(v); t_vocab = time.time() - t0
    gv.set_option(W.WP_OPT_STAGE_TIMING, 1)
    gv.encode(text); gv.encode(text)
    st = gv.stats()
    print("vocab %7d lines (create %.2f s): device %.1f ms = decode %.1f sa %.1f scan %.1f walk %.1f; n=%d rounds=%d" % (
        len(v), t_vocab, st["ms_total"], st["ms_decode"], st["ms_sa"], st["ms_scan"], st["ms_walk"], st["n_total"], st["rounds"]), flush=True)
