"""Soak run on the GPU box (not collected by pytest): thousands of adversarial random cases through the
default path (text-only layout, pruned, depth capped), the reference layout, the multi-context entry point
and the fast path, each against the CPU oracle.  usage: python tests/soak_gpu.py [cases] [seed]"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: F401,E402
import oracle_lib as O  # noqa: E402
import wordpiece_amd as W  # noqa: E402


def make_case(rng, k):
    kind = k % 8
    if kind == 0:    # tiny alphabet, long repetitive text, long tokens (streams far beyond the 32-bit key)
        alpha, tok_max, text_len = "ab", 40, rng.randint(100, 20000)
    elif kind == 1:  # skewed alphabet: one very frequent symbol (1-2 bit code) and rare ones (12-bit codes)
        alpha, tok_max, text_len = "a" * 40 + "bcdefghijklmnopqrstuvwxyzABCDEFGH", 12, rng.randint(50, 8000)
    elif kind == 2:  # spacing chars inside tokens (soft), punctuation, CJK
        alpha, tok_max, text_len = "ab-, .c中文▁", 6, rng.randint(0, 3000)
    elif kind == 3:  # words with shared long prefixes
        alpha, tok_max, text_len = "abc ", 30, rng.randint(200, 30000)
    elif kind == 4:  # wide alphabet (> 255 symbols: u32 symbols, split code)
        alpha, tok_max, text_len = "".join(chr(c) for c in range(0x400, 0x400 + 300)) + " ab", 8, rng.randint(50, 5000)
    elif kind == 5:  # big case: full-size radix tiles and digit bytes (n > 2^21)
        alpha, tok_max, text_len = "etaoinshr dlu ", 20, rng.randint(2_200_000, 5_200_000)  # (> 2^22: LDS-window rank store, ranks inside its first pass)
    elif kind == 6:  # invalid UTF-8 sprinkled in
        alpha, tok_max, text_len = "ab c", 10, rng.randint(10, 2000)
    else:
        alpha, tok_max, text_len = "abcdefgh ij", 18, rng.randint(0, 6000)
    nt = rng.randint(1, 40)
    vocab = set()
    base = "".join(rng.choice(alpha.replace(" ", "")) for _ in range(tok_max)) if kind in (0, 3) else None
    while len(vocab) < nt:
        ln = rng.randint(1, tok_max)
        if base is not None and rng.random() < 0.6:
            w = base[:ln]  # prefixes of one long word: many long tokens with one key
        else:
            w = "".join(rng.choice(alpha) for _ in range(ln))
        if not w.strip():
            continue
        if rng.random() < 0.4:
            w = "##" + w
        vocab.add(w)
    vocab = sorted(vocab)
    rng.shuffle(vocab)
    if rng.random() < 0.4:
        vocab.append("[UNK]")
    if kind == 3 or kind == 0:
        words = [w.lstrip("#") for w in vocab if w != "[UNK]"] + [base]
        parts = []
        n = 0
        while n < text_len:
            w = rng.choice(words)
            cut = rng.randint(1, len(w))
            piece = w[:cut] + (rng.choice(words)[:rng.randint(0, 8)] if rng.random() < 0.5 else "")
            parts.append(piece)
            n += len(piece) + 1
        text = " ".join(parts)
    else:
        text = "".join(rng.choice(alpha) for _ in range(text_len))
    tb = text.encode("utf8")
    if kind == 6:
        bb = bytearray(tb)
        for _ in range(rng.randint(1, 6)):
            bb.insert(rng.randint(0, len(bb)), rng.choice([0xff, 0xc0, 0x80, 0xe2, 0xf0]))
        tb = bytes(bb)
    return tb, vocab


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    done = bad = 0
    k = 0
    while done < cases:
        text, vocab = make_case(rng, k)
        k += 1
        try:
            ov = O.Vocab(vocab)
        except O.OracleError:
            continue
        exp = ov.encode(text, threads=8 if len(text) > 1_000_000 else 1)
        gv = W.Vocab(vocab)
        got = gv.encode(text)
        st = gv.stats()
        checks = [("default", got)]
        if done % 3 == 0:
            g2 = W.Vocab(vocab)
            g2.set_option(W.WP_OPT_VOCAB_IN_S, 1)
            checks.append(("vocab_in_s", g2.encode(text)))
        if done % 5 == 0:
            checks.append(("multi", gv.encode_multi(text, [0, 0])))
        for name, ids in checks:
            if not np.array_equal(ids, exp):
                bad += 1
                print("MISMATCH", name, "case", k - 1, "kind", (k - 1) % 8, repr(text[:120]), vocab[:12], st, flush=True)
        fexp = ov.fast_encode(text, threads=8 if len(text) > 1_000_000 else 1)
        if not np.array_equal(gv.fast_encode(text), fexp):
            bad += 1
            print("MISMATCH fast case", k - 1, repr(text[:120]), vocab[:12], flush=True)
        done += 1
        if done % 200 == 0:
            print("soak: %d cases, %d mismatches (last: kind %d, n=%d, rounds=%d, needed=%d)"
                  % (done, bad, (k - 1) % 8, st.get("n_total", 0), st.get("rounds", 0), st.get("needed_after_round0", 0)), flush=True)
    print("SOAK DONE: %d cases, %d mismatches" % (done, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
