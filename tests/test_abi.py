"""CPU checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the host vocab logic matches the oracle, and compute entry points fail loudly
without a GPU (no fallback)."""
import os
import re

import pytest

import oracle_lib as O
import wordpiece_amd as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(W.LIB_PATH):
        from wordpiece_amd import build
        build.build()


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "wordpiece_amd.h")).read()
    declared = set(re.findall(r"\b(wp_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(W.ABI_SYMBOLS)
    L = W.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_vocab_classification_matches_oracle():
    words = ["a", "##a", "[UNK]", "[a]", "##[a]", "...", ".", "##..", "a.", "[]", "#", "###", "####", "[ ]", ", ",
             "aé中", "##中", "a b", "[CLS", "CLS]", b"\xffa", b"##\xffb"]
    gv, ov = W.Vocab(words), O.Vocab(words)
    assert len(gv) == ov.size and gv.unk_id == ov.unk_id == 2
    for i in range(len(words)):
        assert gv.token_flags(i) == ov.flags(i), words[i]
        assert gv.token_len(i) == len(ov.word(i)), words[i]


def test_empty_word_error_message():
    for bad in (["a", "##"], ["a", ""], [b"\xff"]):
        with pytest.raises(W.WordPieceError, match="Vocab word is empty"):
            W.Vocab(bad)


def test_vocab_from_file_getline_semantics(tmp_path):
    p = tmp_path / "vocab.txt"
    p.write_bytes(b"[UNK]\na\r\n##b\nlast")
    v = W.Vocab(file=str(p))
    assert len(v) == 4 and v.unk_id == 0
    assert v.token_len(1) == 2  # the CR stays in the token (utils.cpp:123-137)


def test_empty_text_needs_no_device():
    assert W.linear.encode("", ["a"]) == []


def test_no_cpu_fallback():
    if W.lib().wp_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(W.WordPieceError, match="no HIP device"):
        W.linear.encode("ab", ["a", "##b"])


def test_shard_bounds_cut_at_whitespace():
    data = b"alpha beta\tgamma\ndelta epsilon zeta"
    for ws in (1, 2, 3, 4, 8):
        b = W.shard_bounds(data, ws)
        assert b[0][0] == 0 and b[-1][1] == len(data)
        for (s0, e0), (s1, e1) in zip(b, b[1:]):
            assert e0 == s1
            assert e0 == len(data) or data[e0] in b" \t\n"


def test_file_entry_points_report_missing_files(tmp_path):
    """linear.cpp:337-374: a missing text file throws (Boost's mapped_file there, mmap here); a missing
    vocab file is an empty vocabulary (std::ifstream reads nothing, utils.cpp:123-137) and an output
    file that cannot be created is an error.  All before the device is needed."""
    vf = tmp_path / "v.txt"
    vf.write_text("a\nb\n")
    with pytest.raises(W.WordPieceError, match="cannot open"):
        W.linear.encode(str(tmp_path / "missing.txt"), str(vf))
    with pytest.raises(W.WordPieceError, match="cannot open"):
        W.linear.encodeExternal(str(vf), str(vf), str(tmp_path / "no_such_dir" / "out.txt"), 1000)
    with pytest.raises(W.WordPieceError, match="memory_limit too small"):
        W.linear.encodeExternal(str(vf), str(vf), str(tmp_path / "out.txt"), 10)
    empty = tmp_path / "empty.txt"
    empty.write_bytes(b"")
    assert W.linear.encode(str(empty), str(tmp_path / "missing_vocab.txt")) == []  # nothing to encode, nothing loaded


def test_stats_struct_matches_header():
    """The ctypes mirror of wp_stats (field order, widths, the array) follows include/wordpiece_amd.h, and the
    option numbers of the Python module are the header's."""
    import ctypes as C
    import re
    hdr = open(os.path.join(ROOT, "include", "wordpiece_amd.h")).read()
    body = hdr[hdr.index("typedef struct {", hdr.index("statistics of the last encode")):hdr.index("} wp_stats;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct {", "").strip()
        if not decl:
            continue
        ctype, names = decl.split(None, 1)
        for name in names.split(","):
            name = name.strip()
            m = re.match(r"(\w+)\[(\d+)\]", name)
            fields.append((m.group(1), ctype, int(m.group(2))) if m else (name, ctype, 0))
    widths = {"int64_t": C.c_int64, "int32_t": C.c_int32, "double": C.c_double}
    mirror = W.Stats._fields_
    assert [f[0] for f in fields] == [f[0] for f in mirror]
    for (name, ctype, count), (_, pytype) in zip(fields, mirror):
        want = widths[ctype] * count if count else widths[ctype]
        assert C.sizeof(pytype) == C.sizeof(want), name
    for name, value in re.findall(r"#define (WP_OPT_\w+) (\d+)", hdr):
        assert getattr(W, name) == int(value), name
