"""Small-input latency: one encode of a short text (a) on a persistent vocab handle, (b) through the
reference-shaped one-shot API (vocab parsed, context made and destroyed per call, as
word_piece::linear::encode(text, vocab) does), for a 5-line and a 29k-line vocabulary."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import wordpiece_amd as W
from wordpiece_amd import synth

def med(f, n=30):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort(); return round(ts[len(ts) // 2] * 1e3, 3)

small_vocab = ["self", "made", "-", "##-", "##made"]
text_big, big_vocab = synth.english_corpus(2_000_000, seed=2, vocab_size=29000)
out = {}
for name, vocab in (("vocab5", small_vocab), ("vocab29k", big_vocab)):
    gv = W.Vocab(vocab)
    gv.encode(b"self-made")
    for label, text in (("9B", b"self-made"), ("1KB", text_big[:1000]), ("100KB", text_big[:100_000]), ("2MB", text_big)):
        out["%s_handle_%s_ms" % (name, label)] = med(lambda: gv.encode(text))
        out["%s_handle_fast_%s_ms" % (name, label)] = med(lambda: gv.fast_encode(text))
    out["%s_oneshot_9B_ms" % name] = med(lambda: W.linear.encode("self-made", vocab), 10)
print(json.dumps(out))
