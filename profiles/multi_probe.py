"""Round structure of config 3 (mixed scripts, alphabet > 255) at a reduced size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from wordpiece_amd import synth
import wordpiece_amd as W
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
text, vocab = synth.multilingual_corpus(nbytes, seed=200, vocab_size=120000)
gv = W.Vocab(vocab)
gv.set_option(W.WP_OPT_STAGE_TIMING, 1)
for it in range(2):
    gv.encode(text)
    st = gv.stats()
    print("multilingual %d bytes, n=%d alphabet=%d bits=%d symbols/key=%d: device %.1f ms (sa %.1f walk %.1f), active per round %s" % (
        len(text), st["n_total"], st["alphabet"], st["symbol_bits"], st["symbols_per_key"], st["ms_total"], st["ms_sa"],
        st["ms_walk"], st["active_per_round"]), flush=True)
