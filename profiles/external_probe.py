"""encodeExternal (linear.cpp:343-374) on a file: wall time per stage-less call, output size, md5."""
import hashlib, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wordpiece_amd import synth
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
batch = int(float(sys.argv[2])) if len(sys.argv) > 2 else 250_000_000
d = tempfile.mkdtemp(dir="/tmp")
parts, vocab = [], None
for k in range(max(1, nbytes // 100_000_000)):
    t, v = synth.english_corpus(min(nbytes, 100_000_000), seed=1)
    vocab = vocab or v
    parts.append(t if k == 0 else t[::-1].replace(b"\n", b" ")[:len(t)])  # cheap variety: reversed copy
    break
text = parts[0]
reps = max(1, nbytes // len(text))
tf, vf, out = os.path.join(d, "t.txt"), os.path.join(d, "v.txt"), os.path.join(d, "ids.txt")
with open(tf, "wb") as f:
    for _ in range(reps):
        f.write(text)
open(vf, "wb").write("\n".join(vocab).encode() + b"\n")
import wordpiece_amd as W
W.linear.encodeExternal(tf, vf, out, 20 * 1_000_000)  # warm-up on nothing big: first batch sizes the arenas
for it in range(2):
    t0 = time.time()
    W.linear.encodeExternal(tf, vf, out, 20 * batch)
    dt = time.time() - t0
    sz = os.path.getsize(out)
    print("encodeExternal: %.0f MB in, %.0f MB of id text out, batches of %.0f MB: %.2f s = %.0f MB/s" % (
        reps * len(text) / 1e6, sz / 1e6, batch / 1e6, dt, reps * len(text) / 1e6 / dt), flush=True)
h = hashlib.md5()
with open(out, "rb") as f:
    for blk in iter(lambda: f.read(1 << 24), b""):
        h.update(blk)
print("md5 of the id file:", h.hexdigest())
for p in (tf, vf, out):
    os.remove(p)
