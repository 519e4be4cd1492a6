"""tests.cpp:259-272 shape: one long lowercase word split into 30000 tokens (a single-word text)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from wordpiece_amd import synth
import wordpiece_amd as W
import numpy as np
for text_len in (1_000_000, 10_000_000):
    for positive in (True, False):
        s, vocab = synth.random_split_case(7 + text_len, text_len, 30000, positive)
        gv = W.Vocab(vocab)
        gv.set_option(W.WP_OPT_STAGE_TIMING, 1)
        gv.encode(s)
        t0 = time.time(); ids = gv.encode(s); dt = time.time() - t0
        st = gv.stats()
        print("len %d positive %s: %d ids, %.1f ms (sa %.1f walk %.1f), rounds %d, longest token %d, anchors %d mode %d" % (
            text_len, positive, len(ids), st["ms_total"], st["ms_sa"], st["ms_walk"], st["rounds"], st["longest_token"],
            st["n_anchors"], st["anchor_mode"]), flush=True)
