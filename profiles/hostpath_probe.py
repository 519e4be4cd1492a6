"""Host-buffer path of wp_linear_encode: where the wall time goes (upload, device, download, Python copy)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wordpiece_amd import synth
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
text, vocab = synth.english_corpus(nbytes, seed=1)
import wordpiece_amd as W
gv = W.Vocab(vocab)
gv.set_option(W.WP_OPT_STAGE_TIMING, 1)
gv.encode(text[:1_000_000])
for it in range(3):
    t0 = time.time(); ids = gv.encode(text); wall = (time.time() - t0) * 1e3
    st = gv.stats()
    print("bytes %d ids %d: wall %.1f ms = h2d %.1f + device %.1f + d2h %.1f + python copy/rest %.1f" % (
        len(text), len(ids), wall, st["ms_h2d"], st["ms_total"], st["ms_d2h"],
        wall - st["ms_h2d"] - st["ms_total"] - st["ms_d2h"]), flush=True)
