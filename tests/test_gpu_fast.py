"""GPU tests (-m gpu) of word_piece::fast on the device (csrc/fast.h, through the C ABI) against the
oracle's restatement of fast.cpp, against the reference's known answers, and against the Linear path
on the device (tests/tests.cpp:80-97: linear == fast on vocabularies without duplicate or
spacing-char-spanning tokens) — an on-device differential check that does not go through the oracle."""
import json
import os
import random
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import wordpiece_amd as W
from wordpiece_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(os.path.abspath(W.__file__))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)["cases"]


def test_reference_vectors_fast():
    for case in _load("reference_tests_cpp.json"):
        text = bytes.fromhex(case["text_hex"])
        vocab = [bytes.fromhex(w) for w in case["vocab_hex"]]
        gv = W.Vocab(vocab)
        got = gv.fast_encode(text).tolist()
        if case["expected"] is not None:  # tests.cpp:80-88
            assert got == case["expected"]
        assert got == gv.encode(text).tolist()  # tests.cpp:90-97: linear == fast, both on the device
        assert got == O.Vocab(vocab).fast_encode(text).tolist()


def test_survey_cases_fast_vs_oracle():
    for case in _load("survey_probed_cases.json"):
        text = bytes.fromhex(case["text_hex"])
        vocab = [bytes.fromhex(w) for w in case["vocab_hex"]]
        assert W.Vocab(vocab).fast_encode(text).tolist() == O.Vocab(vocab).fast_encode(text).tolist(), case["name"]
    # the two Fast outputs SURVEY.md 0.2 recorded from the reference
    assert W.fast.encode("ab-cd", ["ab-cd", "ab", "-", "cd"]) == [1, 2, 3]
    assert W.fast.encode("ab ab", ["ab", "x", "ab"]) == [2, 2]
    assert W.fast.encode("", ["a"]) == [] and W.fast.encode("  \n\t ", ["a"]) == []


def test_fast_fuzz_vs_oracle():
    rng = random.Random(2024)
    alpha = "ab-, .c中文▁"
    done = 0
    while done < int(os.environ.get("WP_FUZZ_FAST", "800")):
        nt = rng.randint(1, 10)
        vocab = set()
        while len(vocab) < nt:
            w = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 5)))
            if rng.random() < 0.4:
                w = "##" + w
            vocab.add(w)
        vocab = sorted(vocab)
        rng.shuffle(vocab)
        if rng.random() < 0.3:
            vocab.append("[UNK]")
        if rng.random() < 0.1:
            vocab.append(vocab[0])
        text = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 80) if done % 10 else rng.randint(500, 6000)))
        try:
            ov = O.Vocab(vocab)
        except O.OracleError:
            continue
        assert W.Vocab(vocab).fast_encode(text).tolist() == ov.fast_encode(text).tolist(), (text[:200], vocab)
        done += 1


def test_random_split_grid_linear_equals_fast_on_device():
    """tests.cpp:219-246, 257-272 (own generator): text_len x parts, positive and negative, up to the
    reference's large single-word shapes with a 30,000-entry vocabulary."""
    k = 0
    for text_len, parts_list in ((10, (2, 7)), (100, (2, 30, 100)), (300, (7, 100)), (5000, (30, 1000)),
                                 (100_000, (30_000,)), (500_000, (30_000,)), (900_000, (30_000,))):
        for parts in parts_list:
            for positive in (True, False):
                s, vocab = synth.random_split_case(9000 + k, text_len, parts, positive)
                k += 1
                if not vocab:
                    continue
                gv = W.Vocab(vocab)
                f = gv.fast_encode(s)
                assert np.array_equal(f, gv.encode(s)), (text_len, parts, positive)
                assert np.array_equal(f, O.Vocab(vocab).fast_encode(s)), (text_len, parts, positive)


def test_fast_english_and_multilingual_vs_oracle_and_linear():
    text, vocab = synth.english_corpus(16_000_000, seed=44)
    gv = W.Vocab(vocab)
    f = gv.fast_encode(text)
    assert np.array_equal(f, O.Vocab(vocab).fast_encode(text, threads=8))
    assert np.array_equal(f, gv.encode(text))  # linear == fast on the device
    text, vocab = synth.multilingual_corpus(8_000_000, seed=45, vocab_size=30000)
    gv = W.Vocab(vocab)
    f = gv.fast_encode(text)
    assert np.array_equal(f, O.Vocab(vocab).fast_encode(text, threads=8))
    # (the multilingual vocabulary holds multi-char CJK tokens: Linear matches across CJK chars, Fast takes
    # a CJK char's segment up to the next spacing char — they may differ there)


def test_fast_long_words_and_blanks():
    vocab = ["a", "##b", "x", "##x", "ab", "##ab", "[UNK]", "中", "##c"]
    ov, gv = O.Vocab(vocab), W.Vocab(vocab)
    for text in (b"ab " * 5 + b"x" * 300_000 + b" ab",            # long word that tokenizes
                 b"ab " + b"x" * 100_000 + b"q" + b"x" * 9 + b" ab",  # long word that fails late: one [UNK]
                 b"a" + b" " * 200_000 + b"ab",                      # long blank run
                 "中".encode() + b"c" * 50_000 + b" ab",              # CJK char + long run: stays with its lane
                 b"x" * 5000 + b"," + b"x" * 5000):
        assert np.array_equal(gv.fast_encode(text), ov.fast_encode(text)), text[:20]
        assert gv.stats()["n_ids"] >= 1


def test_fast_file_external_runner_and_decode(tmp_path):
    text, vocab = synth.english_corpus(2_000_000, seed=46, vocab_size=4000)
    text = text[:700_000] + " привет▁мир 中文 ".encode() + text[700_000:]
    vocab = list(vocab) + ["##文", "中", "при", "##вет", "мир", ", ,"]
    tf, vf, out = tmp_path / "t.txt", tmp_path / "v.txt", tmp_path / "ids.txt"
    tf.write_bytes(text)
    vf.write_bytes("\n".join(vocab).encode() + b"\n")
    exp = O.Vocab(vocab).fast_encode(text).tolist()
    assert W.fast.encode(str(tf), str(vf)) == exp
    W.fast.encodeExternal(str(tf), str(vf), str(out), 2 * 500_000)  # fast.cpp:195: batches of limit/2 bytes
    got = open(out).read()
    assert got == "".join("%d " % i for i in [int(x) for x in got.split(" ") if x])
    # batches are cut where a space starts, so the concatenation equals the unbatched encode
    assert [int(x) for x in got.split(" ") if x] == exp
    r = subprocess.run([os.path.join(PKG, "runner"), "fast", str(tf), str(vf), "0", str(out)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "Total ids %d" % len(exp)
    assert open(out).read() == "".join("%d " % i for i in exp)
    r = subprocess.run([os.path.join(PKG, "runner"), "fast-external", str(tf), str(vf), "0", str(out), "50"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and [int(x) for x in open(out).read().split(" ") if x] == exp
    # fast::decode (fast.cpp:172-187): "##" restored, unknown ids and malformed tokens skipped
    ids = exp[:50] + [-1, len(vocab) + 5, len(vocab) - 1]
    dec = W.fast.decode(str(vf), ids)
    want = [w.encode() for w in (vocab[i] for i in exp[:50])]
    assert dec == want
