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
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Tests are finished" in r.stdout


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
