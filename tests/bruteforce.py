"""Independent brute-force model of the Linear WordPiece result (pure Python, tiny
inputs only): per position scan every eligible token of the position's class and
keep the longest that matches inside S = text·1·vocab, then apply the walk rules
S5-S9 of SURVEY.md §0.1.  Used to cross-check the C oracle on random cases."""


def is_space(c):
    return (c < 256 and (0x09 <= c <= 0x0D or c == 0x20)) or c == 9601


def is_punct(c):
    if c < 256 and (0x21 <= c <= 0x2F or 0x3A <= c <= 0x40 or 0x5B <= c <= 0x60 or 0x7B <= c <= 0x7E):
        return True
    return c in (183, 171, 187, 8249, 8250) or 8208 <= c <= 8248


def is_chinese(c):
    return (0x4E00 <= c <= 0x9FFF or 0x3400 <= c <= 0x4DBF or 0x20000 <= c <= 0x2A6DF
            or 0x2A700 <= c <= 0x2B73F or 0x2B740 <= c <= 0x2B81F or 0x2B820 <= c <= 0x2CEAF
            or 0xF900 <= c <= 0xFAFF or 0x2F800 <= c <= 0x2FA1F)


def is_spacing(c):
    return is_space(c) or is_punct(c) or is_chinese(c)


def decode(b):
    """Strict UTF-8 decoder dropping every byte that does not start a valid sequence."""
    out, i, n = [], 0, len(b)
    while i < n:
        c = b[i]
        if c < 0x80:
            out.append(c)
            i += 1
            continue
        need = 2 if c & 0xE0 == 0xC0 else 3 if c & 0xF0 == 0xE0 else 4 if c & 0xF8 == 0xF0 else 0
        ok = need and i + need <= n and all(b[i + k] & 0xC0 == 0x80 for k in range(1, need))
        if ok:
            cp = c & (0x1F if need == 2 else 0x0F if need == 3 else 0x07)
            for k in range(1, need):
                cp = (cp << 6) | (b[i + k] & 0x3F)
            lo = {2: 0x80, 3: 0x800, 4: 0x10000}[need]
            if cp >= lo and (cp < 0xD800 or 0xDFFF < cp < 0x110000):
                out.append(cp)
                i += need
                continue
        i += 1
    return out


def encode(text, vocab):
    text = text if isinstance(text, (bytes, bytearray)) else text.encode("utf8")
    if len(text) == 0:
        return []
    toks, unk = [], -1
    for i, w in enumerate(vocab):
        w = w if isinstance(w, (bytes, bytearray)) else w.encode("utf8")
        if w == b"[UNK]":
            unk = i
        cps = decode(w)
        prefix, special = True, False
        if len(cps) >= 2 and cps[0] == 35 and cps[1] == 35:
            prefix, cps = False, cps[2:]
        elif len(cps) > 2 and cps[0] == 91 and cps[-1] == 93:
            special = True
        if not cps:
            raise RuntimeError("Vocab word is empty")
        malformed = len(cps) > 1 and all(is_punct(c) or is_space(c) for c in cps)
        toks.append((prefix, special or malformed, cps))
    t = decode(text)
    S = t + [1]
    for _, _, cps in toks:
        S += cps + [1]
    n = len(t)

    def wp(p):
        return p == 0 or is_spacing(t[p]) or is_spacing(t[p - 1])

    def best(p, prefix):
        b, bl = -1, 0
        for i, (pf, bad, cps) in enumerate(toks):
            if pf == prefix and not bad and len(cps) > bl and S[p:p + len(cps)] == cps:
                b, bl = i, len(cps)
        return b, bl

    out, p, tsp = [], 0, 0
    while p != n and is_space(t[p]):
        p += 1
    while p < n:
        b, bl = best(p, wp(p))
        if b != -1:
            tsp += 1
            out.append(b)
            p += bl
            if p != n and p < n and wp(p):
                tsp = 0
        else:
            del out[len(out) - tsp:]
            tsp = 0
            out.append(unk)
            p += 1
            while p != n and p < n and not wp(p):
                p += 1
        while p < n and is_space(t[p]):
            p += 1
    return out
