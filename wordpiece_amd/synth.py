"""Deterministic synthetic corpora and vocabularies.

bert-base-cased `vocab.txt` and enwiki are not available offline, so the bench
and the parity tests use these generators: English-shaped ASCII text (config 2),
mixed en/ru/ja/zh text (config 3) and 512-char words over a prefix-heavy
vocabulary (config 5).  Pure numpy; chunked so that 1 GB fits in host memory.
"""
import numpy as np

_LETTERS = "etaoinshrdlcumwfgypbvkjxqz"
_ALNUM = [chr(c) for c in range(48, 58)] + [chr(c) for c in range(65, 91)] + [chr(c) for c in range(97, 123)]
_PUNCT = list(",.;:!?()'\"-")
_SPECIALS = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]


def _gather_concat(src, starts, lens):
    """Concatenate src[starts[k]:starts[k]+lens[k]] for all k (vectorised)."""
    total = int(lens.sum())
    out_start = np.cumsum(lens) - lens
    idx = np.repeat(starts - out_start, lens) + np.arange(total, dtype=np.int64)
    return src[idx]


def _lexicon(rng, n_words, max_len=18):
    lens = np.clip(rng.poisson(5.5, n_words), 1, max_len).astype(np.int64)
    w = 1.0 / np.arange(1, len(_LETTERS) + 1)
    letters = np.frombuffer(_LETTERS.encode(), dtype=np.uint8)
    flat = letters[rng.choice(len(_LETTERS), size=int(lens.sum()), p=w / w.sum())].copy()
    off = np.cumsum(lens) - lens
    cap = rng.random(n_words) < 0.15
    flat[off[cap]] -= 32
    # de-duplicate words (keeps Zipf ranks meaningful)
    words = [flat[o:o + l].tobytes() for o, l in zip(off, lens)]
    seen, uniq = set(), []
    for x in words:
        if x not in seen:
            seen.add(x)
            uniq.append(x)
    return uniq


def english_vocab(lex, rng, vocab_size=29000, n_top=18000):
    """5 specials + alnum chars + punctuation + ##alnum + top lexicon words + random ## infixes."""
    vocab = list(_SPECIALS) + _ALNUM + _PUNCT + ["##" + c for c in _ALNUM]
    seen = set(vocab)
    for w in lex[:min(n_top, max(0, vocab_size - len(vocab) - 64))]:
        s = w.decode()
        if s not in seen:
            seen.add(s)
            vocab.append(s)
    w = 1.0 / np.arange(1, len(_LETTERS) + 1)
    w /= w.sum()
    while len(vocab) < vocab_size:
        k = int(rng.integers(1, 5))
        s = "##" + "".join(_LETTERS[i] for i in rng.choice(len(_LETTERS), size=k, p=w))
        if s not in seen:
            seen.add(s)
            vocab.append(s)
    return vocab


def english_corpus(n_bytes, seed=0, vocab_size=29000, lexicon_size=200000, chunk=32 << 20, text_seed=None):
    """English-shaped ASCII text of ~n_bytes bytes (cut at a separator) + a BERT-like vocab.

    Zipf(s=1.05) words from a lexicon with Poisson(5.5) lengths; separators
    {' ' x7, ', ', '. ', '\\n'}/10.  Returns (bytes, list[str]).
    text_seed: draw the word sequence from a separate stream (same lexicon and vocab, different
    text: the shards of a multi-GPU run)."""
    rng = np.random.default_rng(seed)
    lex = _lexicon(rng, lexicon_size)
    vocab = english_vocab(lex, rng, vocab_size)
    if text_seed is not None:
        rng = np.random.default_rng([seed, text_seed])
    seps = [b" "] * 7 + [b", ", b". ", b"\n"]
    table = lex + seps
    src = np.frombuffer(b"".join(table), dtype=np.uint8)
    tlen = np.array([len(x) for x in table], dtype=np.int64)
    toff = np.cumsum(tlen) - tlen
    p = 1.0 / np.arange(1, len(lex) + 1) ** 1.05
    cdf = np.cumsum(p / p.sum())
    parts, have = [], 0
    while have < n_bytes:
        want = min(chunk, n_bytes - have)
        nw = int(want / 6.3) + 16
        wid = np.minimum(np.searchsorted(cdf, rng.random(nw)), len(lex) - 1)
        sid = len(lex) + rng.integers(0, len(seps), nw)
        piece = np.empty(2 * nw, dtype=np.int64)
        piece[0::2] = wid
        piece[1::2] = sid
        ends = np.cumsum(tlen[piece])
        k = int(np.searchsorted(ends, want))
        k = min(len(piece), k + (k % 2 == 0) + 1)  # end on a separator
        k -= k % 2
        piece = piece[:max(k, 2)]
        buf = _gather_concat(src, toff[piece], tlen[piece])
        parts.append(buf)
        have += len(buf)
    return np.concatenate(parts).tobytes(), vocab


def _utf8_encode(cps):
    """Vectorised UTF-8 encoder for code points < 0x10000 (numpy uint32 array)."""
    cps = cps.astype(np.uint32)
    nb = np.where(cps < 0x80, 1, np.where(cps < 0x800, 2, 3)).astype(np.int64)
    off = np.cumsum(nb) - nb
    out = np.zeros(int(nb.sum()), dtype=np.uint8)
    m1, m2, m3 = nb == 1, nb == 2, nb == 3
    out[off[m1]] = cps[m1]
    out[off[m2]] = 0xC0 | (cps[m2] >> 6)
    out[off[m2] + 1] = 0x80 | (cps[m2] & 0x3F)
    out[off[m3]] = 0xE0 | (cps[m3] >> 12)
    out[off[m3] + 1] = 0x80 | ((cps[m3] >> 6) & 0x3F)
    out[off[m3] + 2] = 0x80 | (cps[m3] & 0x3F)
    return out


def _script_block(rng, n_bytes, lo, hi, bytes_per_cp, lex_size, spaced=True, mix=None):
    """Words over code points [lo,hi) (optionally mixed with a second range), Zipf-drawn."""
    lens = np.clip(rng.poisson(4.5, lex_size), 1, 12).astype(np.int64)
    flat = rng.integers(lo, hi, int(lens.sum())).astype(np.uint32)
    if mix is not None:
        m = rng.random(len(flat)) < 0.3
        flat[m] = rng.integers(mix[0], mix[1], int(m.sum()))
    off = np.cumsum(lens) - lens
    p = 1.0 / np.arange(1, lex_size + 1) ** 1.05
    cdf = np.cumsum(p / p.sum())
    nw = int(n_bytes / (bytes_per_cp * 4.6 + 1)) + 8
    wid = np.minimum(np.searchsorted(cdf, rng.random(nw)), lex_size - 1)
    sep = np.array([32], dtype=np.uint32)
    src = np.concatenate([flat, sep])
    if spaced:
        piece_start = np.empty(2 * nw, dtype=np.int64)
        piece_len = np.empty(2 * nw, dtype=np.int64)
        piece_start[0::2], piece_len[0::2] = off[wid], lens[wid]
        piece_start[1::2], piece_len[1::2] = len(flat), 1
    else:
        piece_start, piece_len = off[wid], lens[wid]
    cps = _gather_concat(src, piece_start, piece_len)
    words = [flat[o:o + l] for o, l in zip(off[:20000], lens[:20000])]
    return cps, words


def multilingual_corpus(n_bytes, seed=0, vocab_size=120000):
    """Config 3: four equal blocks en / ru (2-byte) / ja (kana+CJK, 3-byte) / zh (CJK, 3-byte)."""
    rng = np.random.default_rng(seed)
    q = n_bytes // 4
    en_text, en_vocab = english_corpus(q, seed=seed + 1, vocab_size=min(29000, vocab_size // 4))
    blocks = [np.frombuffer(en_text, dtype=np.uint8)]
    vocab, seen = list(en_vocab), set(en_vocab)
    specs = [(0x0410, 0x0450, 2, 60000, True, None),        # ru
             (0x3040, 0x3100, 3, 60000, True, (0x4E00, 0x6000)),  # ja: kana mixed with CJK
             (0x4E00, 0x6000, 3, 60000, False, None)]       # zh: every char its own word
    for lo, hi, bpc, lex, spaced, mix in specs:
        cps, words = _script_block(rng, q, lo, hi, bpc, lex, spaced, mix)
        blocks.append(_utf8_encode(cps))
        blocks.append(np.array([10], dtype=np.uint8))
        used = np.unique(cps)
        for c in used[used > 32]:
            for s in (chr(int(c)), "##" + chr(int(c))):
                if s not in seen:
                    seen.add(s)
                    vocab.append(s)
        for w in words:
            if len(vocab) >= vocab_size:
                break
            s = "".join(chr(int(c)) for c in w)
            if s not in seen:
                seen.add(s)
                vocab.append(s)
            s2 = "##" + s[: max(1, len(s) // 2)]
            if s2 not in seen and len(vocab) < vocab_size:
                seen.add(s2)
                vocab.append(s2)
    return np.concatenate(blocks).tobytes(), vocab


def deep_prefix_corpus(n_bytes, seed=0, word_len=512, n_stems=128, suffix_stems=16, suffix_len=128,
                       words_seed=None):
    """Config 5: words = stem_k[:m] + stem_j[:word_len-m]; vocab = every proper prefix of every stem.
    words_seed: draw the words from a separate stream (same stems and vocab, different text)."""
    rng = np.random.default_rng(seed)
    stems = rng.integers(97, 123, size=(n_stems, word_len)).astype(np.uint8)
    if words_seed is not None:
        rng = np.random.default_rng([seed, words_seed])
    vocab = ["[UNK]"] + ["##" + chr(c) for c in range(97, 123)]
    for k in range(n_stems):
        s = stems[k].tobytes().decode()
        vocab += [s[:m] for m in range(1, word_len + 1)]
    for k in range(suffix_stems):
        s = stems[k].tobytes().decode()
        vocab += ["##" + s[:m] for m in range(2, suffix_len + 1)]
    vocab = list(dict.fromkeys(vocab))
    nw = max(1, n_bytes // (word_len + 1))
    a = rng.integers(0, n_stems, nw)
    b = rng.integers(0, n_stems, nw)
    m = rng.integers(64, word_len, nw)
    out = np.empty((nw, word_len + 1), dtype=np.uint8)
    col = np.arange(word_len)[None, :]
    out[:, :word_len] = np.where(col < m[:, None], stems[a][np.arange(nw)[:, None], col],
                                 stems[b][np.arange(nw)[:, None], np.maximum(col - m[:, None], 0)])
    out[:, word_len] = 32
    return out.tobytes(), vocab


def random_split_case(seed, text_len, parts, positive=True):
    """The reference's testRandomSplit-style case (tests.cpp:99-135), own generator."""
    rng = np.random.default_rng(seed)
    s = rng.integers(97, 123, text_len).astype(np.uint8).tobytes().decode()
    borders = {text_len}
    while len(borders) < min(parts, text_len):
        borders.add(int(rng.integers(1, text_len)))
    res, start = set(), 0
    for b in sorted(borders):
        if start == 0:
            res.add(s[start:b])
        res.add("##" + s[start:b])
        start = b
    vocab = sorted(res)
    if not positive:
        vocab = vocab[1:]
    return s.encode(), vocab


# ---- large shards: ~100 MB chunks from worker processes -------------------------------------------------
def _gen_chunk(args):
    kind, nbytes, seed, k, vocab_size = args
    if kind == "english":
        text, vocab = english_corpus(nbytes, seed=seed, vocab_size=vocab_size, text_seed=(1000 * seed + k) if k else None)
    elif kind == "multilingual":
        text, vocab = multilingual_corpus(nbytes, seed=seed + 7 * k, vocab_size=vocab_size)
    else:
        text, vocab = deep_prefix_corpus(nbytes, seed=seed, words_seed=k if k else None)
    if not text.endswith((b" ", b"\n")):
        text += b"\n"
    return text, (vocab if k == 0 else None)


def parallel_corpus(kind, nbytes, seed=2, vocab_size=29000, rank=0, workers=None):
    """A shard of ~nbytes: chunks of ~100 MB generated by worker processes (plain `python -c` children
    that import this module only: they never touch the GPU and do not depend on how the parent was
    started), the vocabulary from chunk 0 of rank 0 (english / deep: one lexicon and vocabulary for
    every chunk and rank, a different word sequence per chunk; multilingual: chunk 0's vocabulary).
    kind: "english" | "multilingual" | "deep".  Returns (bytes, vocab)."""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    nbytes = int(nbytes)
    nchunks = max(1, int(round(nbytes / 100e6)))
    if nchunks == 1:
        if kind == "english":
            return english_corpus(nbytes, seed=seed, vocab_size=vocab_size, text_seed=rank if rank > 0 else None)
        return _gen_chunk((kind, nbytes, seed, 0, vocab_size))
    jobs = [(kind, nbytes // nchunks, seed, (64 * rank + k) if (rank or k) else 0, vocab_size) for k in range(nchunks)]
    if rank > 0:  # chunk 0 of rank 0 defines the vocabulary: regenerate it for the other ranks
        jobs.insert(0, (kind, 4_000_000 if kind != "multilingual" else nbytes // nchunks, seed, 0, vocab_size))
    workers = workers or min(len(jobs), max(2, (os.cpu_count() or 8) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))), 16)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json; sys.path.insert(0, %r); from wordpiece_amd import synth; a = json.loads(sys.argv[1]); "
            "t, v = synth._gen_chunk(tuple(a)); open(sys.argv[2], 'wb').write(t); "
            "json.dump([w if isinstance(w, str) else w.decode('utf8') for w in v], open(sys.argv[2] + '.vocab', 'w')) if v is not None else None" % root)
    parts = [None] * len(jobs)
    with tempfile.TemporaryDirectory(prefix="wp_corpus_") as tmp:
        running = {}
        nxt = 0
        while nxt < len(jobs) or running:
            while nxt < len(jobs) and len(running) < workers:
                path = os.path.join(tmp, "chunk%d.bin" % nxt)
                running[nxt] = (subprocess.Popen([sys.executable, "-c", code, json.dumps(list(jobs[nxt])), path]), path)
                nxt += 1
            done = [k for k, (pr, _) in running.items() if pr.poll() is not None]
            if not done:
                next(iter(running.values()))[0].wait(timeout=600)
                continue
            for k in done:
                pr, path = running.pop(k)
                if pr.returncode != 0:
                    for other, _ in running.values():
                        other.kill()
                    raise RuntimeError("corpus worker %d failed (exit %d)" % (k, pr.returncode))
                with open(path, "rb") as f:
                    text = f.read()
                vocab = None
                if os.path.exists(path + ".vocab"):
                    with open(path + ".vocab") as f:
                        vocab = json.load(f)
                os.remove(path)
                parts[k] = (text, vocab)
    vocab = parts[0][1]
    if rank > 0:
        parts = parts[1:]
    return b"".join(p[0] for p in parts), vocab
