"""Kernel-level view of config 5 (deep prefixes) at a reduced size: python profiles/deep_probe.py <bytes>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wordpiece_amd import synth
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
text, vocab = synth.deep_prefix_corpus(nbytes, seed=300)
import wordpiece_amd as W
gv = W.Vocab(vocab)
gv.set_option(W.WP_OPT_STAGE_TIMING, 1)
for it in range(3):
    ids = gv.encode(text)
    st = gv.stats()
    print("deep %d bytes: device %.1f ms (sa %.1f), rounds %d, active per round %s" % (
        len(text), st["ms_total"], st["ms_sa"], st["rounds"], st["active_per_round"]), flush=True)
