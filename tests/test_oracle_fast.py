"""CPU tests pinning the oracle's restatement of word_piece::fast (oracle/wp_oracle.c, fast.cpp:19-158):
the reference's own known-answer vectors (tests/tests.cpp:80-88 asserts them for Fast as well as for
Linear), its Linear == Fast differential shapes (tests.cpp:90-97, 219-246), the two Fast outputs
SURVEY.md section 0.2 recorded from the reference (Q1, Q9), and an independent pure-Python model."""
import json
import os
import random

import numpy as np

import bruteforce as B
import oracle_lib as O
from wordpiece_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)["cases"]


def fast_model(text, vocab):
    """fast.cpp:19-108 transliterated into plain Python (dict lookups of tuples), single worker."""
    text = B.decode(text if isinstance(text, (bytes, bytearray)) else text.encode("utf8"))
    if not text:
        return []
    maps, unk, max_len = ({}, {}), -1, 0
    for i, w in enumerate(vocab):
        w = w if isinstance(w, (bytes, bytearray)) else w.encode("utf8")
        if w == b"[UNK]":
            unk = i
        cps, prefix, special = B.decode(w), True, False
        if len(cps) >= 2 and cps[0] == 35 and cps[1] == 35:
            prefix, cps = False, cps[2:]
        elif len(cps) > 2 and cps[0] == 91 and cps[-1] == 93:
            special = True
        malformed = len(cps) > 1 and all(B.is_punct(c) or B.is_space(c) for c in cps)
        if special or malformed:
            continue
        max_len = max(max_len, len(cps))
        maps[0 if prefix else 1][tuple(cps)] = i
    max_len = min(max_len, len(text))
    n = len(text)

    def wp(i):
        return i == 0 or B.is_spacing(text[i]) or B.is_spacing(text[i - 1])

    out, p, since = [], 0, 0
    while p != n and B.is_space(text[p]):
        p += 1
    while p != n:
        wl = 1
        if not B.is_punct(text[p]):
            while wl < min(max_len, n - p) and not B.is_spacing(text[p + wl]):
                wl += 1
        m = maps[0 if wp(p) else 1]
        ln = wl
        while ln > 0 and tuple(text[p:p + ln]) not in m:
            ln -= 1
        if ln > 0:
            since += 1
            out.append(m[tuple(text[p:p + ln])])
            p += ln
            if p != n and wp(p):
                since = 0
        else:
            del out[len(out) - since:]
            since = 0
            out.append(unk)
            p += wl
            while p != n and not wp(p):
                p += 1
        while p != n and B.is_space(text[p]):
            p += 1
    return out


def test_reference_vectors_hold_for_fast():
    for case in _load("reference_tests_cpp.json"):
        text = bytes.fromhex(case["text_hex"])
        vocab = [bytes.fromhex(w) for w in case["vocab_hex"]]
        ov = O.Vocab(vocab)
        got = ov.fast_encode(text).tolist()
        if case["expected"] is not None:  # tests.cpp:80-88: check(text, vocab, expected) asserts linear AND fast
            assert got == case["expected"]
        else:  # tests.cpp:90-97: linear == fast
            assert got == ov.encode(text).tolist()
        assert got == fast_model(text, vocab)


def test_survey_recorded_fast_outputs():
    """Q1 "ab-cd" {ab-cd,ab,-,cd}: Linear [0], Fast [1,2,3]; Q9 "ab ab" {ab,x,ab}: Fast [2,2]."""
    assert O.Vocab(["ab-cd", "ab", "-", "cd"]).fast_encode("ab-cd").tolist() == [1, 2, 3]
    assert O.Vocab(["ab", "x", "ab"]).fast_encode("ab ab").tolist() == [2, 2]
    assert len(O.Vocab(["a"]).fast_encode("")) == 0 and len(O.Vocab(["a"]).fast_encode("  \n\t ")) == 0


def test_random_split_grid_linear_equals_fast():
    """tests.cpp:219-246, 257-258 (own generator): Linear == Fast on split vocabularies, positive and negative."""
    k = 0
    for text_len in (10, 35, 100, 300, 1000, 5000, 100_000):
        for parts in (2, 7, 30, 100, 3000):
            for positive in (True, False):
                s, vocab = synth.random_split_case(5000 + k, text_len, min(parts, text_len), positive)
                k += 1
                if not vocab:
                    continue
                ov = O.Vocab(vocab)
                assert np.array_equal(ov.fast_encode(s), ov.encode(s)), (text_len, parts, positive)


def test_fast_against_python_model_adversarial():
    rng = random.Random(77)
    alpha = "ab-, .c中"
    done = 0
    while done < 1500:
        nt = rng.randint(1, 8)
        vocab = set()
        while len(vocab) < nt:
            w = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 4)))
            if rng.random() < 0.4:
                w = "##" + w
            vocab.add(w)
        vocab = sorted(vocab)
        rng.shuffle(vocab)
        if rng.random() < 0.3:
            vocab.append("[UNK]")
        if rng.random() < 0.1:
            vocab.append(vocab[0])  # duplicate line: the later one wins (fast.cpp:34)
        text = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 60)))
        try:
            ov = O.Vocab(vocab)
        except O.OracleError:
            continue
        assert ov.fast_encode(text).tolist() == fast_model(text, vocab), (text, vocab)
        done += 1


def test_fast_chunked_equals_single_worker():
    text, vocab = synth.english_corpus(6_000_000, seed=8, vocab_size=3000)
    ov = O.Vocab(vocab)
    a = ov.fast_encode(text)
    assert np.array_equal(a, ov.fast_encode(text, threads=4))
    assert np.array_equal(a, ov.encode(text, threads=4))  # a sane vocabulary: Linear == Fast
