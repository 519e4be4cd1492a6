"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Every stage is diffed in full-depth mode (SA, rank and LCP are then unique and must be
bit-identical to the oracle's); the default depth-capped mode must give identical ids."""
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O
import wordpiece_amd as W
from wordpiece_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)["cases"]


def _combine(d, vocab_lens, which):
    """The reference's merge of the left and right scan answers (linear.cpp:234-250) per SA slot."""
    n = d["n"]
    left = d["best_left_" + which]
    right = d["best_right_" + which][::-1]  # right arrays are in scan order: result[i] <-> slot n-1-i
    lens = np.asarray(list(vocab_lens) + [0], dtype=np.int64)
    lx, ly = lens[left], lens[right]
    both = (left != -1) & (right != -1)
    out = np.where(both, np.where(lx > ly, left, right), np.maximum(left, right))
    assert len(out) == n
    return out.astype(np.int32)


def check_all_stages(text, vocab, label=""):
    text = text if isinstance(text, bytes) else text.encode("utf8")
    ov = O.Vocab(vocab)
    d = ov.encode_debug(text)
    gv = W.Vocab(vocab)
    gv.set_option(W.WP_OPT_FULL_DEPTH, 1)
    gv.set_option(W.WP_OPT_KEEP_DEBUG, 1)  # keeps the raw code points for debug_fetch(6)
    ids = gv.encode(text)
    n = d["n"]
    if len(text) and d["n_text"] > 0:
        cps = gv.debug_fetch(6, d["n_text"])
        assert np.array_equal(cps, d["S"][:d["n_text"]]), label + " code points"
        sym = gv.debug_fetch(0, n)
        uniq = np.unique(d["S"])
        assert np.array_equal(sym, np.searchsorted(uniq, d["S"]) + 1), label + " dense symbols"
        assert np.array_equal(gv.debug_fetch(1, n), d["SA"]), label + " suffix array"
        assert np.array_equal(gv.debug_fetch(2, n), d["rank"]), label + " rank"
        assert np.array_equal(gv.debug_fetch(3, n), d["lcp"]), label + " lcp"
        lens = [ov_len for ov_len in (O.lib().wpo_vocab_token_len(ov._h, i) for i in range(ov.size))]
        assert np.array_equal(gv.debug_fetch(4, n), _combine(d, lens, "prefix")), label + " best prefix"
        assert np.array_equal(gv.debug_fetch(5, n), _combine(d, lens, "suffix")), label + " best suffix"
    assert np.array_equal(ids, d["ids"]), label + " ids (full depth)"
    gv2 = W.Vocab(vocab)
    assert np.array_equal(gv2.encode(text), d["ids"]), label + " ids (depth capped)"
    return gv2.stats()


@pytest.mark.parametrize("case", _load("reference_tests_cpp.json"))
def test_reference_tests_cpp_vectors(case):
    text = bytes.fromhex(case["text_hex"])
    vocab = [bytes.fromhex(w) for w in case["vocab_hex"]]
    if case["expected"] is not None:
        assert W.Vocab(vocab).encode(text).tolist() == case["expected"]
    check_all_stages(text, vocab)


@pytest.mark.parametrize("case", _load("survey_probed_cases.json"), ids=lambda c: c["name"])
def test_survey_probed_cases(case):
    text = bytes.fromhex(case["text_hex"])
    vocab = [bytes.fromhex(w) for w in case["vocab_hex"]]
    assert W.Vocab(vocab).encode(text).tolist() == case["expected"]
    check_all_stages(text, vocab)


def test_errors_match_reference():
    with pytest.raises(W.WordPieceError, match="Vocab word is empty"):
        W.Vocab(["a", "##"])
    assert W.linear.encode("", ["a"]) == []


def test_random_small_adversarial():
    rng = random.Random(4321)
    alpha = "ab-, .c"
    done = 0
    total = int(os.environ.get("WP_FUZZ_SMALL", "300"))
    while done < total:
        nt = rng.randint(1, 8)
        vocab = set()
        while len(vocab) < nt:
            w = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 4)))
            if rng.random() < 0.4:
                w = "##" + w
            vocab.add(w)
        vocab = sorted(vocab)
        rng.shuffle(vocab)
        # (every 10th case is long enough for several doubling rounds and repeated substrings)
        text = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 40) if done % 10 else rng.randint(200, 3000)))
        try:
            O.Vocab(vocab)
        except O.OracleError:
            continue
        check_all_stages(text, vocab, label=repr((text[:80], vocab)))
        done += 1


def test_random_split_grid():
    """tests.cpp:219-246 testRandomSplit (own generator): positive and negative vocabularies."""
    k = 0
    for text_len in (10, 35, 100, 300, 1000, 5000):
        for parts in (2, 7, 30, 100):
            for positive in (True, False):
                s, vocab = synth.random_split_case(1000 + k, text_len, parts, positive)
                k += 1
                if vocab:
                    check_all_stages(s, vocab, label="split %d/%d" % (text_len, parts))


def test_invalid_utf8_and_multibyte():
    rng = random.Random(7)
    pool = [b"a", b"b", b" ", b"\xd0\xbf", b"\xe4\xb8\xad", b"\xf0\x9f\x98\x80", b"\xff", b"\xc0\x80", b"\xed\xa0\x80",
            b"\xe2\x96", b"\x80", b"\xe2\x96\x81", b"\xc3", b"\xf0\x9f", b",", b"\xe6\x96\x87"]
    vocab = ["a", "b", "##a", "##b", "п", "##п", "中", "文", "ab", "##ab", ",", "😀", "[UNK]"]
    for _ in range(60):
        text = b"".join(rng.choice(pool) for _ in range(rng.randint(1, 200)))
        check_all_stages(text, vocab, label=repr(text))


def test_english_1mb_all_stages():
    text, vocab = synth.english_corpus(1_000_000, seed=3, vocab_size=5000)
    st = check_all_stages(text, vocab, "english 1MB")
    assert st["n_ids"] > 100000


def test_multilingual_all_stages():
    text, vocab = synth.multilingual_corpus(600_000, seed=5, vocab_size=20000)
    check_all_stages(text, vocab, "multilingual")


def test_deep_prefix_all_stages():
    text, vocab = synth.deep_prefix_corpus(300_000, seed=9, word_len=128, n_stems=16, suffix_stems=4, suffix_len=32)
    check_all_stages(text, vocab, "deep prefixes")


def test_duplicate_vocab_lines_force_full_depth():
    v = W.Vocab(["ab", "x", "ab"])
    assert v.encode("ab ab").tolist() == [0, 0]
    assert v.stats()["full_depth"] == 1


def test_deep_prefix_8mb_ids_both_refinements():
    """Many doubling rounds over a large list (the reference's layout), and the same ids from one trie walk + one
    segmented sort (the default layout)."""
    text, vocab = synth.deep_prefix_corpus(8_000_000, seed=14)
    gv = W.Vocab(vocab)
    gv.set_option(W.WP_OPT_VOCAB_IN_S, 1)  # (the doubling rounds: the default layout resolves its needed groups along the token trie)
    ids = gv.encode(text)
    assert gv.stats()["rounds"] >= 4
    exp = _oracle_ids_fast(text, vocab)
    assert np.array_equal(ids, exp)
    # default layout: one trie walk + one segmented sort whatever the depth of the vocabulary
    gv = W.Vocab(vocab)
    assert np.array_equal(gv.encode(text), exp)
    st = gv.stats()
    assert st["trie_refine"] == 1 and st["rounds"] == 2 and st["vocab_in_s"] == 0


def _cover_ids(text, vocab):
    gv = W.Vocab(vocab)
    gv.set_option(W.WP_OPT_COVER_ANCHORS, 1)
    ids = gv.encode(text)
    st = gv.stats()
    assert st["anchor_mode"] == 1 or len(text) == 0 or st["n_text"] == 0
    return ids


def test_cover_anchors_forced_on_golden_and_random():
    """WP_OPT_COVER_ANCHORS: the walk's start positions derived from the matches give the same ids."""
    for name in ("reference_tests_cpp.json", "survey_probed_cases.json"):
        for case in _load(name):
            text = bytes.fromhex(case["text_hex"])
            vocab = [bytes.fromhex(w) for w in case["vocab_hex"]]
            exp = case["expected"] if case["expected"] is not None else O.Vocab(vocab).encode(text).tolist()
            assert _cover_ids(text, vocab).tolist() == exp, case.get("name", name)
    rng = random.Random(99)
    alpha = "ab-, .c中"
    done = 0
    while done < 300:
        vocab = set()
        while len(vocab) < rng.randint(1, 8):
            w = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 4)))
            vocab.add("##" + w if rng.random() < 0.4 else w)
        vocab = sorted(vocab)
        text = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 60)))
        try:
            ov = O.Vocab(vocab)
        except O.OracleError:
            continue
        assert _cover_ids(text, vocab).tolist() == ov.encode(text).tolist(), repr((text, vocab))
        done += 1
    text, vocab = synth.english_corpus(2_000_000, seed=41, vocab_size=6000)
    assert np.array_equal(_cover_ids(text, vocab), _oracle_ids_fast(text, vocab))


def test_soft_spacing_chars_switch_to_cover_anchors():
    """CJK text with multi-char CJK tokens: every spacing char is "soft", the class rule finds no
    anchors, and the encoder must switch to the coverage rule by itself (a single lane would
    otherwise walk the whole text)."""
    rng = np.random.default_rng(5)
    chars = [chr(c) for c in range(0x4E00, 0x4E00 + 300)]
    words = ["".join(rng.choice(chars, size=int(k))) for k in rng.integers(1, 5, size=3000)]
    vocab = ["[UNK]"] + chars[:280] + ["##" + c for c in chars[:280]] + list(dict.fromkeys(words[:1500]))
    text = "".join(words[i] for i in rng.integers(0, len(words), size=60000)).encode("utf8")
    gv = W.Vocab(vocab)
    ids = gv.encode(text)
    st = gv.stats()
    assert st["anchor_mode"] == 1 and st["n_anchors"] > st["n_text"] // 64
    assert np.array_equal(ids, O.Vocab(vocab).encode(text))
    # mixed: an English block (dense class-rule anchors) in front of the CJK block
    en, en_vocab = synth.english_corpus(300_000, seed=12, vocab_size=3000)
    vocab2 = en_vocab + [w for w in vocab if w not in set(en_vocab)]
    text2 = en + b"\n" + text
    gv2 = W.Vocab(vocab2)
    ids2 = gv2.encode(text2)
    assert gv2.stats()["anchor_mode"] == 1
    assert np.array_equal(ids2, O.Vocab(vocab2).encode(text2))
    # plain English stays on the class rule
    gv3 = W.Vocab(en_vocab)
    gv3.encode(en)
    assert gv3.stats()["anchor_mode"] == 0


def test_single_word_text_random_split_300k():
    """tests.cpp:259-272 shape: one long lowercase word cut into many tokens; positive (every piece
    in the vocab) and negative (first piece missing: the whole word is one [UNK]-less -1)."""
    for positive in (True, False):
        s, vocab = synth.random_split_case(77, 300_000, 3000, positive)
        gv = W.Vocab(vocab)
        ids = gv.encode(s)
        st = gv.stats()
        assert st["anchor_mode"] == 2 and st["n_anchors"] == 1  # one word, walked by pointer doubling
        exp = _oracle_ids_fast(s, vocab)
        assert np.array_equal(ids, exp)
        assert (len(ids) == 3000) if positive else (ids.tolist() == [-1])
    # long words between ordinary ones: the [UNK] skip jumps over tiles without a word-prefix position
    rng = np.random.default_rng(8)
    words = []
    for k in range(40):
        words.append(rng.integers(97, 123, int(rng.integers(1, 30_000))).astype(np.uint8).tobytes())
        words.append(b"ab" if k % 3 else b"zz,")
    text = b" ".join(words)
    vocab = ["ab", "##b", "a", ",", "zz"]
    assert np.array_equal(W.Vocab(vocab).encode(text), O.Vocab(vocab).encode(text))
    assert np.array_equal(_cover_ids(text, vocab), O.Vocab(vocab).encode(text))
    # base64-like blobs inside ordinary text, BERT-like vocab (every alphanumeric has a ## piece): the
    # blobs tokenize into ~100 k pieces each; one blob holds a byte no piece covers and becomes [UNK]
    en, vocab = synth.english_corpus(300_000, seed=23, vocab_size=4000)
    alnum = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)
    blob = lambda n: alnum[rng.integers(0, len(alnum), n)].tobytes()
    text = (en[:100_000] + b" " + blob(150_000) + b" " + en[100_000:200_000] + b"\n" + blob(40_000) + b"_" + blob(5)
            + b" " + blob(3000) + b"\x01" + blob(3000) + b" " + en[200_000:] + b" " + blob(70_000))
    gv = W.Vocab(vocab)
    ids = gv.encode(text)
    assert gv.stats()["anchor_mode"] == 2
    assert np.array_equal(ids, _oracle_ids_fast(text, vocab))


def test_long_whitespace_runs():
    """Leading / inner / trailing whitespace runs far longer than a tile: no lane may step through them
    (class rule: return at the first hard space; coverage rule: per-tile table of non-space positions)."""
    en, vocab = synth.english_corpus(200_000, seed=19, vocab_size=3000)
    text = b" " * 300_000 + en[:100_000] + b" \n\t" * 150_000 + en[100_000:] + b"\n" * 200_000
    exp = _oracle_ids_fast(text, vocab)
    gv = W.Vocab(vocab)
    assert np.array_equal(gv.encode(text), exp) and gv.stats()["anchor_mode"] == 0
    assert np.array_equal(_cover_ids(text, vocab), exp)          # coverage rule forced
    assert W.Vocab(vocab).encode(b" " * 1_000_000).size == 0      # nothing but blanks
    # a vocabulary with a soft space (a token that contains one): stepping paths
    vocab2 = vocab + ["of the", "##s of"]
    text2 = b"  " * 5000 + en[:50_000] + b" " * 20_000 + en[50_000:100_000]
    assert np.array_equal(W.Vocab(vocab2).encode(text2), O.Vocab(vocab2).encode(text2))
    assert np.array_equal(_cover_ids(text2, vocab2), O.Vocab(vocab2).encode(text2))


def test_fuzz_long_runs_all_anchor_modes():
    """Random vocabularies (with and without spacing chars inside tokens) over texts that contain long
    words, long blank runs and CJK stretches: every walk variant (class rule, coverage rule, long-word
    doubling, [UNK] skips) against the oracle."""
    rng = random.Random(20260)
    letters = "abcdefgh"
    modes = set()
    for case in range(int(os.environ.get("WP_FUZZ_CASES", "120"))):
        soft = case % 3 == 0
        vocab = set()
        for _ in range(rng.randint(3, 25)):
            w = "".join(rng.choice(letters + (",中" if rng.random() < 0.3 else "")) for _ in range(rng.randint(1, 5)))
            if soft and rng.random() < 0.3:
                w = w[:1] + rng.choice([" ", ",", "中"]) + w[1:]
            vocab.add(("##" if rng.random() < 0.4 else "") + w)
        for c in letters[:rng.randint(0, 8)]:
            vocab.add(c)
            vocab.add("##" + c)
        if rng.random() < 0.5:
            vocab.add("[UNK]")
        vocab = sorted(vocab)
        rng.shuffle(vocab)
        try:
            ov = O.Vocab(vocab)
        except O.OracleError:
            continue
        parts = []
        for _ in range(rng.randint(3, 12)):
            kind = rng.random()
            if kind < 0.3:
                parts.append("".join(rng.choice(letters) for _ in range(rng.randint(2100, 5000))))  # long word
            elif kind < 0.45:
                parts.append(rng.choice([" ", "\n", " \t"]) * rng.randint(2100, 4000))               # long blank run
            elif kind < 0.55:
                parts.append("".join(rng.choice("中文字") for _ in range(rng.randint(100, 3000))))
            else:
                parts.append(" ".join("".join(rng.choice(letters + ",") for _ in range(rng.randint(1, 9)))
                                      for _ in range(rng.randint(5, 200))))
            parts.append(rng.choice([" ", "", ",", "\n"]))
        text = "".join(parts).encode("utf8")
        exp = ov.encode(text)
        gv = W.Vocab(vocab)
        got = gv.encode(text)
        modes.add(gv.stats()["anchor_mode"])
        assert np.array_equal(got, exp), (case, vocab)
        assert np.array_equal(_cover_ids(text, vocab), exp), (case, vocab, "coverage rule")
    assert modes == {0, 1, 2} or modes == {1, 2}


def test_periodic_text_full_depth_all_stages():
    """Highly repetitive text: two giant groups that stay tied for log2(n) rounds (the large-group
    path in every round), full depth — SA, rank, LCP and ids bit for bit."""
    text = ("ab" * 60_000 + " " + "abc" * 30_000 + " " + "a" * 50_000).encode()
    vocab = ["ab", "##ab", "abc", "##abc", "a", "##a", "##b", "##c", "[UNK]"]
    st = check_all_stages(text, vocab, "periodic")
    assert st["n_ids"] > 0
    gv = W.Vocab(vocab)
    gv.set_option(W.WP_OPT_FULL_DEPTH, 1)
    gv.encode(text)
    assert gv.stats()["rounds"] >= 12


def test_kasai_kernel_gives_same_lcp():
    text, vocab = synth.english_corpus(300_000, seed=8, vocab_size=3000)
    d = O.Vocab(vocab).encode_debug(text)
    gv = W.Vocab(vocab)
    gv.set_option(W.WP_OPT_LCP_KASAI, 1)
    ids = gv.encode(text)
    assert np.array_equal(gv.debug_fetch(3, d["n"]), d["lcp"])
    assert np.array_equal(ids, d["ids"])


def test_english_16mb_ids_and_properties():
    """BASELINE-size properties: ids equal the oracle's, and the depth-capped SA is a permutation
    sorted by the first sorted_depth symbols."""
    text, vocab = synth.english_corpus(16_000_000, seed=21)
    O.use_libsais(True)
    try:
        exp = O.Vocab(vocab).encode(text, threads=8)
    finally:
        O.use_libsais(False)
    gv = W.Vocab(vocab)
    gv.set_option(W.WP_OPT_KEEP_DEBUG, 1)  # keeps the suffix array for debug_fetch(1)
    ids = gv.encode(text)
    assert np.array_equal(ids, exp)
    st = gv.stats()
    n = st["n_total"]
    sa = gv.debug_fetch(1, n).astype(np.int64)
    assert np.array_equal(np.sort(sa), np.arange(n))
    sym = gv.debug_fetch(0, n).astype(np.int64)
    depth = min(st["sorted_depth"], 64)
    pad = np.concatenate([sym, np.zeros(depth, dtype=np.int64)])
    idx = np.random.default_rng(0).integers(0, n - 1, 200000)
    a, b = sa[idx], sa[idx + 1]
    for k in range(depth):  # lexicographic compare of the first `depth` symbols, sampled
        ca, cb = pad[a + k], pad[b + k]
        assert np.all(ca <= cb)
        keep = ca == cb
        a, b = a[keep], b[keep]
        if len(a) == 0:
            break


def _oracle_ids_fast(text, vocab):
    O.use_libsais(True)
    try:
        return O.Vocab(vocab).encode(text, threads=os.cpu_count() or 8)
    finally:
        O.use_libsais(False)


def test_multilingual_32mb_ids():
    """configs[2]-shaped (mixed en/ru/ja/zh, multibyte, alphabet > 255 -> u32 symbols)."""
    text, vocab = synth.multilingual_corpus(32_000_000, seed=12, vocab_size=60000)
    gv = W.Vocab(vocab)
    ids = gv.encode(text)
    assert gv.stats()["alphabet"] > 255
    assert np.array_equal(ids, _oracle_ids_fast(text, vocab))


def test_deep_prefix_24mb_ids():
    """configs[4]-shaped: 512-char words, every stem prefix is a token (stack depth up to 512)."""
    text, vocab = synth.deep_prefix_corpus(24_000_000, seed=13)
    gv = W.Vocab(vocab)
    ids = gv.encode(text)
    st = gv.stats()
    # (the groups that carry a long token's key are resolved along the token trie: one walk + one segmented sort,
    # however deep the vocabulary — the doubling rounds took 80 + rounds here)
    assert st["longest_token"] == 512 and st["rounds"] == 2 and st["trie_refine"] == 1
    assert 0 < st["needed_after_round0"] < st["n_total"] // 50
    exp = _oracle_ids_fast(text, vocab)
    assert np.array_equal(ids, exp)
    # the same text through the reference's layout: prefix doubling until the depth exceeds the longest token
    gv = W.Vocab(vocab)
    gv.set_option(W.WP_OPT_VOCAB_IN_S, 1)
    assert np.array_equal(gv.encode(text), exp)
    st = gv.stats()
    assert st["trie_refine"] == 0 and st["rounds"] >= 7


def _ids_both_layouts(text, vocab, label=""):
    """Default layout (S = text . 1, vocabulary through the per-handle structure) and the reference's
    S = text . 1 . vocab layout (WP_OPT_VOCAB_IN_S), both depth capped, against the oracle."""
    exp = O.Vocab(vocab).encode(text, threads=8) if len(text) > 2_000_000 else O.Vocab(vocab).encode(text)
    a = W.Vocab(vocab)
    ids = a.encode(text)
    sa = a.stats()
    b = W.Vocab(vocab)
    b.set_option(W.WP_OPT_VOCAB_IN_S, 1)
    ids_b = b.encode(text)
    sb = b.stats()
    assert np.array_equal(ids, exp), label + " text-only layout"
    assert np.array_equal(ids_b, exp), label + " vocab in S"
    assert sb["vocab_in_s"] == 1 or len(text) == 0  # (empty input: the path is not entered, linear.cpp:323-325)
    return sa, sb


def test_vocab_structure_layout_equals_reference_layout():
    """f2: S without the vocabulary.  symbols_n == n_text + 1, same ids as with the vocabulary in S."""
    text, vocab = synth.english_corpus(4_000_000, seed=51, vocab_size=8000)
    sa, sb = _ids_both_layouts(text, vocab, "english")
    assert sa["vocab_in_s"] == 0 and sa["n_total"] == sa["n_text"] + 1 and sb["n_total"] > sb["n_text"] + 1000
    assert 0 <= sa["needed_after_round0"] < sa["n_total"] // 20
    text, vocab = synth.multilingual_corpus(3_000_000, seed=52, vocab_size=20000)
    sa, _ = _ids_both_layouts(text, vocab, "multilingual")
    assert sa["vocab_in_s"] == 0 and sa["alphabet"] > 255
    text, vocab = synth.deep_prefix_corpus(3_000_000, seed=53)
    sa, sb = _ids_both_layouts(text, vocab, "deep")
    assert sa["vocab_in_s"] == 0 and sa["rounds"] == 2 and sb["rounds"] >= 5 and sb["n_total"] - sa["n_total"] > 10_000_000
    for case in _load("reference_tests_cpp.json") + _load("survey_probed_cases.json"):
        t = bytes.fromhex(case["text_hex"])
        vc = [bytes.fromhex(w) for w in case["vocab_hex"]]
        _ids_both_layouts(t, vc, case.get("name", "vector"))
    rng = random.Random(606)
    alpha = "ab-, .c中"
    done = 0
    while done < 400:
        nt = rng.randint(1, 9)
        vocab = set()
        while len(vocab) < nt:
            w = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 24 if done % 7 == 0 else 5)))
            if rng.random() < 0.4:
                w = "##" + w
            vocab.add(w)
        vocab = sorted(vocab)
        rng.shuffle(vocab)
        # (long tokens and long repetitive texts: tokens whose code stream exceeds the 63-bit key)
        text = "".join(rng.choice(alpha if done % 3 else "ab") for _ in range(rng.randint(0, 60) if done % 5 else rng.randint(300, 4000)))
        if done % 11 == 0 and vocab:
            text += " " + vocab[0].lstrip("#") * 3
        try:
            O.Vocab(vocab)
        except O.OracleError:
            continue
        _ids_both_layouts(text, vocab, repr((text[:60], vocab)))
        done += 1


def test_low_code_points_use_the_reference_layout():
    """U+0000 / U+0001 in the text or in a token sort around the separator (code point 1, linear.cpp:92):
    those inputs keep S = text . 1 . vocab."""
    vocab = ["a", "##b", "ab", "[UNK]"]
    for text in (b"ab a\x01b ab", b"ab\x00 ab"):
        gv = W.Vocab(vocab)
        assert np.array_equal(gv.encode(text), O.Vocab(vocab).encode(text))
        assert gv.stats()["vocab_in_s"] == 1
    vocab = ["a", "##b", "a\x01", "ab"]
    gv = W.Vocab(vocab)
    assert np.array_equal(gv.encode(b"ab a\x01 ab"), O.Vocab(vocab).encode(b"ab a\x01 ab"))
    assert gv.stats()["vocab_in_s"] == 1


def test_staged_and_sparse_id_output_agree():
    """The walk leaves its ids either as per-workgroup lists (default where one kernel produces all ids) or in
    the per-position array compacted afterwards (WP_OPT_SPARSE_EMIT; long words; coverage anchors).  Same ids,
    Linear and fast, against the oracle: words that fail after several tokens (roll-back), words of more than
    kStageIds tokens (ids beyond the LDS stage), empty and all-blank inputs, long words (sparse by necessity)."""
    rng = random.Random(99)
    vocab = ["[UNK]", "a", "b", "##a", "##b", "ab", "##ab", "abab", "##c", "c", ",", "##abababab"]
    texts = [b"", b"   ", b"a", b"ab" * 40 + b"d " + b"abc" * 9 + b" , c,ab", (b"ab" * 3 + b" ") * 3000 + b"abd " * 100]
    for _ in range(40):
        words = []
        for _ in range(rng.randint(1, 400)):
            words.append("".join(rng.choice("abcd,") for _ in range(rng.choice([1, 2, 3, 5, 9, 30]))))
        texts.append(" ".join(words).encode())
    texts.append(b"ab" * 5000 + b" a b " + b"ba" * 3000 + b"d")  # long words: pointer doubling, sparse output
    big, big_vocab = synth.english_corpus(3_000_000, seed=77, vocab_size=8000)
    cases = [(t, vocab) for t in texts] + [(big, big_vocab)]
    for text, vc in cases:
        ov = O.Vocab(vc)
        exp = ov.encode(text, threads=8) if len(text) > 1_000_000 else ov.encode(text)
        exp_fast = ov.fast_encode(text)
        a, b = W.Vocab(vc), W.Vocab(vc)
        b.set_option(W.WP_OPT_SPARSE_EMIT, 1)
        ia, ib = a.encode(text), b.encode(text)
        label = repr(text[:50])
        assert np.array_equal(ia, exp) and np.array_equal(ib, exp), label
        if len(text) and not text.isspace():
            long_words = max(len(w) for w in text.split()) > 4000
            assert a.stats()["staged_emit"] == (0 if long_words else 1), label
            assert b.stats()["staged_emit"] == 0, label
        fa, fb = a.fast_encode(text), b.fast_encode(text)
        assert np.array_equal(fa, exp_fast) and np.array_equal(fb, exp_fast), label


def test_needed_list_outgrows_its_room_and_the_encode_retries():
    """The active list of round 0 is sized from a guess (an eighth of the text, or what the handle's last encode needed).
    A periodic text whose every suffix shares its key with long tokens puts the whole text on the list: the device keeps
    the list empty, reports the length it wanted, and the encode runs again with room — same ids, list_retries == 1 once."""
    words = [b"ab" * 40, b"ab" * 33 + b"c", b"ba" * 25]
    rng = random.Random(77)
    text = b" ".join(rng.choice(words) for _ in range(70_000))
    vocab = ["[UNK]"] + ["ab" * k for k in (1, 2, 5, 9, 14, 20, 33, 40)] + ["##" + "ab" * k for k in (1, 3, 7, 12, 21)] + \
            ["ba" * k for k in (1, 4, 11, 25)] + ["##c", "##b", "##a", "a", "b"]
    exp = _oracle_ids_fast(text, vocab)  # (the reference's libsais: a periodic text is the slow case of a doubling sorter)
    gv = W.Vocab(vocab)
    ids = gv.encode(text)
    st = gv.stats()
    assert np.array_equal(ids, exp)
    assert st["list_retries"] == 1 and st["needed_after_round0"] > st["n_total"] // 2
    assert np.array_equal(gv.encode(text), exp) and gv.stats()["list_retries"] == 0  # (the handle remembers)


def test_lean_walk_paths():
    """The lean walk (walk.h) answers the common step itself and hands the rest to the generic step: tokens of 15
    symbols or more (the landing position lies outside its 16 class bytes), runs of blanks behind a token when
    spacing chars occur inside tokens ("soft": no stop at the first blank), words without a token ([UNK] and the
    roll-back), a text that ends in blanks, texts of fewer than 16 symbols.  Every case against the oracle."""
    rng = random.Random(77)
    letters = "abcdefghij"
    stems = ["".join(rng.choice(letters) for _ in range(k)) for k in (1, 2, 3, 5, 8, 13, 14, 15, 16, 17, 24, 40)]
    vocab = ["[UNK]"] + stems + ["##" + s for s in stems] + list(letters[:6]) + ["##" + c for c in letters[:6]]
    soft = vocab + ["a-b", "##c-d", "e.f", "x-", "##-"]  # '-' and '.' inside tokens: soft spacing chars
    for voc, name in ((vocab, "hard"), (soft, "soft")):
        ov, gv = O.Vocab(voc), W.Vocab(voc)
        for trial in range(120):
            parts = []
            for _ in range(rng.randint(1, 60)):
                w = "".join(rng.choice(stems) for _ in range(rng.randint(1, 3)))
                if rng.random() < 0.15:
                    w += rng.choice(["q", "-", ".", "a-b", "zz"])  # pieces the vocabulary may not cover
                parts.append(w)
                parts.append(rng.choice([" ", " ", "  ", "   ", "\n", " \t ", "-", ". "]))
            text = "".join(parts)
            if trial % 3 == 0:
                text = text.rstrip() + " " * rng.randint(1, 20)
            if trial % 7 == 0:
                text = text[:rng.randint(0, 15)]
            t = text.encode("utf8")
            assert np.array_equal(gv.encode(t), ov.encode(t)), (name, trial, text[:80])
        # the same through the coverage rule
        gv.set_option(W.WP_OPT_COVER_ANCHORS, 1)
        for trial in range(40):
            text = " ".join("".join(rng.choice(stems + ["-", "q"]) for _ in range(rng.randint(1, 4))) for _ in range(200))
            t = text.encode("utf8")
            assert np.array_equal(gv.encode(t), ov.encode(t)), (name, "cover", trial)
