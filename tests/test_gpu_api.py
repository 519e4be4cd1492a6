"""GPU tests of the boundary itself: the C++ API (include/word_piece.hpp) through its own test
executable, the runner CLI with the reference runner's positionals (tests/runner.cpp:13-65), the
file entry points and encodeExternal's batch rule / output format (linear.cpp:343-374)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import wordpiece_amd as W
from wordpiece_amd import synth

pytestmark = pytest.mark.gpu
PKG = os.path.dirname(os.path.abspath(W.__file__))


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    d = tmp_path_factory.mktemp("corpus")
    text, vocab = synth.english_corpus(3_000_000, seed=31, vocab_size=4000)
    # sprinkle multi-byte and invalid bytes, U+2581 spaces and CRLF to exercise the batch cut rule
    text = text[:1_000_000] + "é▁ж中\xff ".encode("latin-1", "ignore") + " привет▁мир ".encode() + text[1_000_000:]
    tf, vf = d / "text.txt", d / "vocab.txt"
    tf.write_bytes(text)
    vf.write_bytes("\n".join(vocab).encode() + b"\n")
    return str(tf), str(vf), text, vocab, d


def test_cpp_api_known_answers():
    exe = os.path.join(PKG, "test_word_piece")
    assert os.path.exists(exe), "run `python -m wordpiece_amd.build`"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Tests are finished" in r.stdout
    # the reference's own acceptance grid at its real size (tests.cpp:257-265): ~29,900 linear == fast checks
    import re
    m = re.search(r"Passed (\d+) checks", r.stdout)
    assert m and int(m.group(1)) >= 29000, r.stdout


def test_encode_files_equals_oracle(corpus):
    tf, vf, text, vocab, _ = corpus
    ids = W.linear.encode(tf, vf)
    assert ids == O.Vocab(vocab).encode(text).tolist()


def test_runner_linear_and_output_format(corpus):
    tf, vf, text, vocab, d = corpus
    out = str(d / "ids.txt")
    r = subprocess.run([os.path.join(PKG, "runner"), "linear", tf, vf, "0", out], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    exp = O.Vocab(vocab).encode(text)
    assert r.stdout.strip() == "Total ids %d" % len(exp)
    # utils.cpp:30-35: decimal ids, each followed by one space, no newline
    assert open(out).read() == "".join("%d " % i for i in exp)


def test_encode_external_batches(corpus):
    """linear.cpp:343-374: batches of memory_limit/20 bytes grown to the next space; every batch is
    encoded on its own (invalid/cut bytes included), ids appended as text."""
    tf, vf, text, vocab, d = corpus
    out = str(d / "ids_ext.txt")
    limit = 20 * 400_000
    W.linear.encodeExternal(tf, vf, out, limit)
    ov = O.Vocab(vocab)
    L = O.lib()
    L.wpo_chars_to_utf8.restype = C.c_uint32
    L.wpo_chars_to_utf8.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_uint64)]

    def starts_with_space(ptr, avail):  # utf8.cpp:92-96 on the byte at `ptr`
        ln = C.c_uint64()
        cp = L.wpo_chars_to_utf8(text[ptr:ptr + 4] + b"\0\0\0\0", avail, C.byref(ln))
        return bool(L.wpo_is_space(cp))

    exp, pos, size, mb = [], 0, len(text), limit // 20
    while size > 0:
        batch = size
        if size > mb:
            batch = mb
            while batch < size and not starts_with_space(pos + batch - 1, size - batch):
                batch += 1
        exp += ov.encode(text[pos:pos + batch]).tolist()
        pos += batch
        size -= batch
    assert open(out).read() == "".join("%d " % i for i in exp)


def test_encode_external_id_text_all_widths(tmp_path):
    """The id text is written by a device kernel (format.h): ids of 1..6 digits and -1 (no [UNK] in
    the vocab), several batches, compared with the reference's `fout << id << ' '` format."""
    rng = np.random.default_rng(3)
    letters = "abcdefghijklmnopqrstuvwxyz"

    def word(k):  # distinct lowercase words, 4..5 letters
        s = ""
        for _ in range(5):
            s += letters[k % 26]
            k //= 26
        return s

    vocab = [word(k) for k in range(120_000)]           # ids 0..119999: one to six digits
    pick = np.concatenate([rng.integers(0, 10, 500), rng.integers(10, 1000, 2000),
                           rng.integers(1000, 120_000, 60_000)])
    rng.shuffle(pick)
    words = [vocab[i] for i in pick]
    for k in range(0, len(words), 50):
        words[k] = "zzzzzz"                              # not in the vocab: -1
    text = " ".join(words).encode()
    tf, vf, out = tmp_path / "t.txt", tmp_path / "v.txt", tmp_path / "ids.txt"
    tf.write_bytes(text)
    vf.write_bytes("\n".join(vocab).encode() + b"\n")
    W.linear.encodeExternal(str(tf), str(vf), str(out), 20 * 100_000)
    got = open(out).read()
    ids = [int(x) for x in got.split(" ") if x]
    assert got == "".join("%d " % i for i in ids)        # exact format
    assert -1 in ids and max(ids) >= 100_000 and min(i for i in ids if i >= 0) < 10
    # batches are cut at spaces here, so the concatenation equals the unbatched encode
    assert ids == W.Vocab(vocab).encode(text).tolist()
    assert ids == O.Vocab(vocab).encode(text).tolist()


def test_encode_tensor_on_device(corpus):
    """Text and ids stay in HBM (wp_linear_encode_device behind a torch tensor view)."""
    import torch
    _, _, text, vocab, _ = corpus
    gv = W.Vocab(vocab)
    exp = O.Vocab(vocab).encode(text[:500_003])
    t = torch.frombuffer(bytearray(text[:500_003]), dtype=torch.uint8).cuda()  # odd length: staged and padded
    ids = gv.encode_tensor(t)
    assert ids.is_cuda and ids.dtype == torch.int32 and np.array_equal(ids.cpu().numpy(), exp)
    view = gv.encode_tensor(t, copy=False)
    assert np.array_equal(view.cpu().numpy(), exp)
    assert gv.encode_tensor(torch.zeros(0, dtype=torch.uint8, device="cuda")).numel() == 0


def test_encode_multi_same_gpu_contexts(corpus):
    """wp_linear_encode_multi: the sharded path behind the C ABI (one host thread + context per entry of
    the device list, whitespace cuts balanced by code points, ids downloaded in shard order).  Several
    contexts on the one GPU of this box exercise everything but the second device."""
    _, _, text, vocab, _ = corpus
    exp = O.Vocab(vocab).encode(text)
    gv = W.Vocab(vocab)
    for devs in ([0], [0, 0], [0, 0, 0]):
        ids = gv.encode_multi(text, devs)
        assert np.array_equal(ids, exp), devs
        st = gv.stats()
        assert st["n_devices"] == len(devs) and st["n_bytes"] == len(text) and st["n_ids"] == len(exp)
    assert np.array_equal(gv.encode_multi(text, None), exp)       # all visible GPUs
    assert len(gv.encode_multi(b"", [0, 0])) == 0
    exp_ab = O.Vocab(vocab).encode(b"ab")
    for _ in range(25):  # empty shards: no worker thread, no context for a shard without bytes
        assert np.array_equal(gv.encode_multi(b"ab", [0, 0, 0]), exp_ab)
        assert np.array_equal(gv.encode_multi(b"ab cd", [0, 0, 0, 0]), O.Vocab(vocab).encode(b"ab cd"))
    # mixed scripts: the cut positions follow the code points, not the bytes
    mixed = ("привет мир " * 40000).encode() + text[:400_000] + ("中文 分词 " * 30000).encode()
    assert np.array_equal(gv.encode_multi(mixed, [0, 0]), O.Vocab(vocab).encode(mixed))
    # WP_OPT_DEVICES routes the plain host entry point (and word_piece::linear::encode) through the same path
    gv.set_option(W.WP_OPT_DEVICES, -1)
    big = text + b" " + text
    assert np.array_equal(gv.encode(big), O.Vocab(vocab).encode(big))
    with pytest.raises(W.WordPieceError, match="no such HIP device"):
        gv.encode_multi(text, [0, 99])


def test_reserve_and_pinned_id_blocks(corpus):
    """wp_reserve pre-sizes the arenas; the id buffers come from a pool of pinned blocks that wp_free
    feeds (a block is reused by the next call once its numpy view is gone)."""
    _, _, text, vocab, _ = corpus
    gv = W.Vocab(vocab)
    gv.reserve(len(text))
    exp = O.Vocab(vocab).encode(text)
    seen = set()
    for _ in range(6):  # (blocks go back to the pool when their numpy view dies: a handful of addresses at most)
        a = gv.encode(text)
        seen.add(a.ctypes.data)
        assert np.array_equal(a, exp)
        del a
    assert len(seen) <= 4
    assert gv.stats()["ms_host_total"] > 0


def test_arena_guard_zones_intact(corpus):
    """WP_OPT_ARENA_GUARD: a guard zone behind every arena allocation, checked after the encode — no
    kernel of the path writes outside the buffer it was given (small- and full-tile radix
    configurations, large-group path, coverage anchors, long words)."""
    _, _, text, vocab, _ = corpus
    cases = [(text, vocab)]
    cases.append(synth.deep_prefix_corpus(3_000_000, seed=5))          # large groups: side-stream radix sorts
    cases.append(synth.multilingual_corpus(3_000_000, seed=6, vocab_size=6000))  # wide symbols, coverage anchors
    cases.append((b"ab " * 5 + b"x" * 300_000 + b" ab", ["a", "##b", "x", "##x"]))  # long word
    for t, vc in cases:
        for full in (0, 1):
            gv = W.Vocab(vc)
            gv.set_option(W.WP_OPT_ARENA_GUARD, 1)
            gv.set_option(W.WP_OPT_FULL_DEPTH, full)
            ids = gv.encode(t)
            assert gv.stats()["guard_zones"] > 50
            if len(t) <= 3_100_000 and full == 0:
                assert np.array_equal(ids, O.Vocab(vc).encode(t))


def test_encode_batch_pipeline_equals_single_calls(corpus):
    """wp_linear_encode_batch: a sequence of texts through one handle with uploads / kernels / downloads of neighbouring
    texts overlapped (second text buffer, id staging): per text the same ids as a call of its own, in order, for
    texts of different sizes, empty texts and texts without ids."""
    _, _, text, vocab, _ = corpus
    gv = W.Vocab(vocab)
    ov = O.Vocab(vocab)
    parts = [text, b"", text[:1000], b"   ", text[5000:900_000], "привет мир ".encode() * 1000, text, text[:1_500_000]]
    for _ in range(2):  # (second round: buffers are reused)
        outs = gv.encode_batch(parts)
        assert len(outs) == len(parts)
        for t, o in zip(parts, outs):
            assert np.array_equal(o, ov.encode(t))
    st = gv.stats()
    assert st["n_bytes"] == sum(len(p) for p in parts)
    assert gv.encode_batch([]) == []
    # the same pipeline with callbacks (wp_linear_encode_stream): ids arrive in order, valid during the callback
    got = []
    gv.encode_stream(iter(parts), lambda i, ids: got.append((i, ids.copy())))
    assert [g[0] for g in got] == list(range(len(parts)))
    for t, (_, o) in zip(parts, got):
        assert np.array_equal(o, ov.encode(t))
    gv.encode_stream([], lambda i, ids: got.append(None))
    assert len(got) == len(parts)


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two ranks are started by bench.py itself, rank 0's line
    says n_gpus 2 and names the id gather (gloo rehearsal on this one-GPU box: the ranks share the device)."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WP_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mb", "8", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["id_gather"].startswith("gloo") and out["value"] > 0


def test_calls_keep_the_current_device_and_trim_releases(corpus):
    """Every entry point of the C ABI leaves the calling thread's current HIP device as it found it (a handle on
    another GPU, wp_linear_encode_multi and wp_vocab_destroy all switch devices inside), and wp_trim gives the
    arenas back.  With one GPU the handle lives on device 0 as well; on a multi-GPU box it is put on the last one."""
    import ctypes as C
    import torch
    _, _, text, vocab, _ = corpus
    W.lib()
    try:  # the HIP runtime the library itself is linked against (by soname: the copy that is already loaded)
        hip = C.CDLL("libamdhip64.so.7")
    except OSError:
        hip = C.CDLL("libamdhip64.so")

    def current():
        d = C.c_int(-1)
        assert hip.hipGetDevice(C.byref(d)) == 0
        return d.value

    ndev = W.lib().wp_device_count()
    assert ndev >= 1
    assert hip.hipSetDevice(0) == 0
    other = ndev - 1
    exp = O.Vocab(vocab).encode(text)
    gv = W.Vocab(vocab, device=other)
    free0 = torch.cuda.mem_get_info(other)[0]
    assert np.array_equal(gv.encode(text), exp) and current() == 0
    assert np.array_equal(gv.fast_encode(text), O.Vocab(vocab).fast_encode(text)) and current() == 0
    assert np.array_equal(gv.encode_multi(text, list(range(ndev)) + [other]), exp) and current() == 0
    gv.reserve(1 << 24)
    assert current() == 0
    used = free0 - torch.cuda.mem_get_info(other)[0]
    assert used > (100 << 20)          # the arenas of a 16 MB reserve
    gv.trim()
    assert current() == 0
    assert free0 - torch.cuda.mem_get_info(other)[0] < used // 4
    assert np.array_equal(gv.encode(text), exp)      # the next encode allocates again
    del gv                                           # wp_vocab_destroy parks the contexts (hipSetDevice inside)
    import gc
    gc.collect()
    assert current() == 0
    W.lib().wp_trim(None)
    assert current() == 0


def test_symbol_limit_is_on_code_points():
    """linear.cpp:104-106 limits total_length (code points + vocab symbols), not bytes: 2.1 GB of ASCII is
    "64bit not implemented"; 2.1 GB of three-byte characters (0.7e9 code points) is encoded."""
    vocab = ["ab", "##c", "中", "文", "[UNK]"]
    gv = W.Vocab(vocab)
    unit = b"abc abd " * 1024
    big = unit * (2_200_000_000 // len(unit) + 1)
    with pytest.raises(W.WordPieceError, match="64bit not implemented"):
        gv.encode(big)
    # the fast path's positions are 32-bit with bit 31 as the skip flag of its sparse walk: 2^31 code points or more
    # are refused as well (fast.cpp has no limit of its own; never silent truncation)
    with pytest.raises(W.WordPieceError, match="64bit not implemented"):
        gv.fast_encode(big)
    del big
    unit = "中文 中 文x ".encode() * 1024
    reps = 2_100_000_000 // len(unit) + 1
    big = unit * reps
    assert len(big) > 2_000_000_000
    ids = gv.encode(big)
    one = gv.encode(unit)
    assert np.array_equal(one, O.Vocab(vocab).encode(unit))
    assert len(ids) == len(one) * reps
    assert np.array_equal(ids[:len(one)], one) and np.array_equal(ids[-len(one):], one)
    assert np.array_equal(ids.reshape(reps, len(one)), np.broadcast_to(one, (reps, len(one))))


def test_id_gather_over_rccl_single_rank(corpus):
    """The collective of bench.py (wordpiece_amd/gather.py) over backend nccl (= RCCL) with one rank on the
    GPU: the RCCL calls of the multi-GPU path run here too (the world_size-2 form runs on gloo in
    tests/test_distributed_gloo.py)."""
    import socket
    import torch
    import torch.distributed as dist
    from wordpiece_amd.gather import IdGather
    _, _, text, vocab, _ = corpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        gv = W.Vocab(vocab, device=0)
        t = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        g = IdGather(dist, 0, 1, dev)
        for _ in range(2):
            ids = gv.encode_tensor(t, copy=False)
            g.step(ids, ids.numel(), before_collective=torch.cuda.current_stream().synchronize)
        dist.barrier()
        assert np.array_equal(g.result(), O.Vocab(vocab).encode(text))
    finally:
        dist.destroy_process_group()


def test_debug_bounds_build_is_clean(tmp_path):
    """libwordpiece_amd_dbg.so (-DWP_DEBUG_BOUNDS): every store / gather whose address comes out of a computed
    table — radix scatter offsets, rank destinations, token ids from the step table, slots of the needed
    list — is range-checked; a violation would make the encode fail with the per-site counts instead of
    faulting.  Run in a child process (the library path is fixed at import time)."""
    dbg = os.path.join(PKG, "libwordpiece_amd_dbg.so")
    assert os.path.exists(dbg), "run `python -m wordpiece_amd.build`"
    script = tmp_path / "dbg_run.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
import numpy as np
import oracle_lib as O, wordpiece_amd as W
from wordpiece_amd import synth
cases = [synth.english_corpus(6_000_000, seed=61, vocab_size=6000),
         synth.multilingual_corpus(3_000_000, seed=62, vocab_size=8000),
         synth.deep_prefix_corpus(3_000_000, seed=63),
         (b"ab " * 7 + b"x" * 200_000 + b" ab", ["a", "##b", "x", "##x", "ab"]),
         (b"ab", ["a", "##b"])]
for text, vocab in cases:
    for opts in ({}, {W.WP_OPT_VOCAB_IN_S: 1}, {W.WP_OPT_FULL_DEPTH: 1}):
        if opts.get(W.WP_OPT_FULL_DEPTH) and len(text) > 4_000_000:
            continue
        gv = W.Vocab(vocab)
        for k, val in opts.items():
            gv.set_option(k, val)
        ids = gv.encode(text)
        assert gv.stats()["reserved0"] == 1, "not the bounds-checking build"
        assert np.array_equal(ids, O.Vocab(vocab).encode(text, threads=8))
        assert np.array_equal(gv.fast_encode(text), O.Vocab(vocab).fast_encode(text, threads=8))
print("DEBUG_BOUNDS_OK")
''' % (os.path.dirname(PKG), os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, WP_LIB=dbg)
    r = subprocess.run([os.sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DEBUG_BOUNDS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_tuning_switches_do_not_change_ids(tmp_path):
    """The environment switches that are left are process-wide defaults of tested behaviours and debugging aids
    (csrc/context.h, EnvOptions): the reference's S layout, the per-position id array, guard zones, no context pool.
    Each in a child process (they are read once per process) on inputs that reach the full-size paths (> 2^22
    symbols), against the oracle."""
    script = tmp_path / "switch_run.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
import numpy as np
import oracle_lib as O, wordpiece_amd as W
from wordpiece_amd import synth
cases = [synth.english_corpus(5_000_000, seed=71, vocab_size=6000),
         synth.deep_prefix_corpus(4_500_000, seed=73),
         (b"ab " * 7 + b"abab" * 300 + b" ab", ["a", "##b", "ab", "##ab", "abab", "[UNK]"])]
for text, vocab in cases:
    exp = O.Vocab(vocab).encode(text, threads=8)
    gv = W.Vocab(vocab)
    for _ in range(2):  # (second call: reused context, cached symbol code)
        assert np.array_equal(gv.encode(text), exp)
print("SWITCH_OK")
''' % (os.path.dirname(PKG), os.path.dirname(os.path.abspath(__file__))))
    combos = [{"WP_VOCAB_IN_S": "1"}, {"WP_SPARSE_EMIT": "1"}, {"WP_ARENA_GUARD": "1", "WP_NO_CONTEXT_POOL": "1"}]
    for combo in combos:
        env = dict(os.environ, **combo)
        r = subprocess.run([os.sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "SWITCH_OK" in r.stdout, str(combo) + r.stdout[-1500:] + r.stderr[-1500:]
