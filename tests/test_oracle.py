"""CPU tests (no GPU): pin the oracle against the reference's own golden vectors,
the survey's recorded outputs, an independent brute-force model, and — when
oracle/_ref was built from /root/reference — the reference's libsais/utf8/utils."""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

import bruteforce
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _load("reference_tests_cpp.json"))
def test_reference_tests_cpp_vectors(case):
    text = bytes.fromhex(case["text_hex"])
    vocab = [bytes.fromhex(w) for w in case["vocab_hex"]]
    got = O.encode(text, vocab).tolist()
    if case["expected"] is not None:
        assert got == case["expected"]
    assert got == bruteforce.encode(text, vocab)


@pytest.mark.parametrize("case", _load("survey_probed_cases.json"), ids=lambda c: c["name"])
def test_survey_probed_cases(case):
    text = bytes.fromhex(case["text_hex"])
    vocab = [bytes.fromhex(w) for w in case["vocab_hex"]]
    assert O.encode(text, vocab).tolist() == case["expected"]


def test_empty_vocab_word_throws():
    for bad in (["a", "##"], ["a", ""], [b"\xff"]):
        with pytest.raises(O.OracleError, match="Vocab word is empty"):
            O.Vocab(bad)


def _random_case(rng):
    alpha = "ab-, .c"
    nt = rng.randint(1, 8)
    vocab = set()
    while len(vocab) < nt:
        w = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 4)))
        if rng.random() < 0.4:
            w = "##" + w
        if w.strip("#") == "" and w.startswith("##") and len(w) == 2:
            continue
        vocab.add(w)
    vocab = sorted(vocab)
    rng.shuffle(vocab)
    text = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 24)))
    return text, vocab


def test_oracle_equals_bruteforce_random():
    rng = random.Random(1234)
    n = 0
    while n < 4000:
        text, vocab = _random_case(rng)
        try:
            exp = bruteforce.encode(text, vocab)
        except RuntimeError:
            continue
        assert O.encode(text, vocab).tolist() == exp, (text, vocab)
        n += 1


def _random_split_case(rng, text_len, parts, positive):
    """tests.cpp:99-135 randomString/randomSplit (our own generator, seeds differ)."""
    s = "".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(text_len))
    borders = {text_len}
    while len(borders) < parts:
        borders.add(rng.randint(1, text_len - 1))
    res, start = set(), 0
    for b in sorted(borders):
        if start == 0:
            res.add(s[start:b])
        res.add("##" + s[start:b])
        start = b
    vocab = sorted(res)
    if not positive:
        vocab = vocab[1:]
    return s, vocab


def test_random_split_grid_vs_bruteforce():
    rng = random.Random(17)
    for text_len in range(10, 120, 5):
        for parts in (2, 3, 5, 9, 17, 33):
            parts = min(parts, text_len)
            for positive in (True, False):
                s, vocab = _random_split_case(rng, text_len, parts, positive)
                if not vocab:
                    continue
                assert O.encode(s, vocab).tolist() == bruteforce.encode(s, vocab)


def test_builtin_sa_is_plain_lexicographic_order():
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = int(rng.integers(1, 60))
        S = rng.integers(0, 4, size=n).astype(np.int32)
        sa = O.suffix_array(S, 4)
        exp = sorted(range(n), key=lambda i: S[i:].tolist())
        assert sa.tolist() == exp


needs_ref = pytest.mark.skipif(not os.path.exists(O.LIBSAIS_REF), reason="oracle/_ref not built")


@needs_ref
def test_builtin_sa_equals_reference_libsais():
    rng = np.random.default_rng(7)
    for n, k in ((1, 3), (2, 2), (1000, 3), (5000, 30), (200000, 5), (300000, 100000)):
        S = rng.integers(0, k, size=n).astype(np.int32)
        S[rng.integers(0, n, size=n // 7 + 1)] = 1
        O.use_libsais(False)
        a = O.suffix_array(S, k)
        assert O.use_libsais(True)
        b = O.suffix_array(S, k)
        O.use_libsais(False)
        assert np.array_equal(a, b)


@needs_ref
def test_encode_same_with_reference_libsais():
    rng = random.Random(99)
    s, vocab = _random_split_case(rng, 20000, 300, True)
    a = O.encode(s, vocab)
    assert O.use_libsais(True)
    b = O.encode(s, vocab)
    O.use_libsais(False)
    assert np.array_equal(a, b) and len(a) > 0


@pytest.mark.skipif(not os.path.exists(O.REFUTILS), reason="oracle/_ref not built")
def test_decode_and_vocab_equal_reference_utils():
    R = C.CDLL(O.REFUTILS)
    R.ref_decode_utf8.restype = C.c_size_t
    R.ref_decode_utf8.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
    R.ref_token.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_int64)]
    for f in ("ref_is_space", "ref_is_punctuation", "ref_is_spacing_char"):
        getattr(R, f).argtypes = [C.c_uint32]
    L = O.lib()
    for c in list(range(0, 0x3100)) + list(range(0x4DB0, 0x4E10)) + list(range(0x9FF0, 0xA010)) + \
            list(range(0xF8F0, 0xFB10)) + list(range(0x1FFF0, 0x20010)) + list(range(0x2A6D0, 0x2A710)) + \
            list(range(0x2B730, 0x2B830)) + list(range(0x2CEA0, 0x2CEC0)) + list(range(0x2F7F0, 0x2FA30)):
        assert bool(L.wpo_is_space(c)) == bool(R.ref_is_space(c)), c
        assert bool(L.wpo_is_punctuation(c)) == bool(R.ref_is_punctuation(c)), c
        assert bool(L.wpo_is_spacing_char(c)) == bool(R.ref_is_spacing_char(c)), c
    rng = random.Random(3)
    pool = [b"a", b"\xd0\xbf", b"\xe4\xb8\xad", b"\xf0\x9f\x98\x80", b"\xff", b"\xc0\x80", b"\xed\xa0\x80",
            b"\xe2\x96", b"\x80", b"\xf4\x90\x80\x80", b" ", b"\xe2\x96\x81", b"\xc3", b"\xf0\x9f", b"\x00", b"\x01"]
    for _ in range(3000):
        b = b"".join(rng.choice(pool) for _ in range(rng.randint(0, 12)))
        out = (C.c_uint32 * (len(b) + 1))()
        n = R.ref_decode_utf8(b, len(b), out)
        mine, _ = O.decode_utf8(b)
        assert mine.tolist() == list(out[:n]), b
        assert mine.tolist() == bruteforce.decode(b), b
    words = ["##", "", "##a", "[UNK]", "[a]", "##[a]", "...", ".", "##..", "a.", "[]", "#", "###", "####",
             "[ ]", ", ", "\xff", "##\xff", "aé中", "##中", "a b", "[CLS", "CLS]"]
    for w in words:
        wb = w.encode("latin-1") if "\xff" in w else w.encode("utf8")
        out = (C.c_uint32 * (len(wb) + 1))()
        ln = C.c_int64()
        fl = R.ref_token(wb, len(wb), out, C.byref(ln))
        if fl < 0:
            with pytest.raises(O.OracleError):
                O.Vocab([wb])
        else:
            v = O.Vocab([wb])
            assert v.flags(0) == fl and v.word(0) == list(out[:ln.value]), w


def test_mt_path_equals_sequential():
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    from wordpiece_amd import synth
    text, vocab = synth.english_corpus(3_000_000, seed=11, vocab_size=3000)
    v = O.Vocab(vocab)
    a = v.encode(text)
    b = v.encode(text, threads=4)
    assert np.array_equal(a, b) and len(a) > 100000
