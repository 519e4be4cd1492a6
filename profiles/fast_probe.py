"""Device-resident rate of word_piece::fast on the bench corpus (100 MB English-shaped shard, 29 k vocab),
next to the Linear path on the same handle; ids of the two compared on the device."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wordpiece_amd as W
from wordpiece_amd import synth

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
text, vocab = synth.english_corpus(int(mb * 1e6), seed=2, vocab_size=29000)
gv = W.Vocab(vocab, device=0)
gv.set_option(W.WP_OPT_STAGE_TIMING, 1)
n = len(text)
d = torch.zeros(n + 32, dtype=torch.uint8, device="cuda")
d[:n] = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
for _ in range(3):
    p, k = gv.fast_encode_device(d.data_ptr(), n)
fast_ids = torch.as_tensor(W.DeviceIds(p, k), device="cuda").clone()
t0 = time.perf_counter()
for _ in range(10):
    gv.fast_encode_device(d.data_ptr(), n)
dt = (time.perf_counter() - t0) / 10
st = gv.stats()
p, k2 = gv.encode_device(d.data_ptr(), n)
lin_ids = torch.as_tensor(W.DeviceIds(p, k2), device="cuda")
print(json.dumps({"bytes": n, "fast_ms": round(dt * 1e3, 3), "fast_MB_per_s": round(n / 1e6 / dt, 1), "n_ids": k,
                  "stage_ms": {a: round(st[a], 3) for a in ("ms_decode", "ms_walk", "ms_total")},
                  "linear_equals_fast_on_device": bool(k == k2 and torch.equal(fast_ids, lin_ids))}))
