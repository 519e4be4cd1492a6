"""ctypes front-end to oracle/liboracle.so — the CPU restatement used as the parity
checker.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the wordpiece_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
LIBSAIS_REF = os.path.join(ODIR, "_ref", "libsais_ref.so")
REFUTILS = os.path.join(ODIR, "_ref", "librefutils.so")


def build():
    subprocess.run(["make", "-s", "-C", ODIR], check=True)


class _Debug(C.Structure):
    _fields_ = [("n_text", C.c_int64), ("n", C.c_int64), ("longest", C.c_int64),
                ("alphabet_size", C.c_uint32),
                ("S", C.POINTER(C.c_int32)), ("SA", C.POINTER(C.c_int32)),
                ("rank", C.POINTER(C.c_int32)), ("lcp", C.POINTER(C.c_int32)),
                ("who", C.POINTER(C.c_int32)),
                ("best_left_prefix", C.POINTER(C.c_int32)), ("best_right_prefix", C.POINTER(C.c_int32)),
                ("best_left_suffix", C.POINTER(C.c_int32)), ("best_right_suffix", C.POINTER(C.c_int32)),
                ("ids", C.POINTER(C.c_int32)), ("n_ids", C.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ODIR, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.wpo_vocab_create.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_void_p)]
        L.wpo_vocab_destroy.argtypes = [C.c_void_p]
        L.wpo_vocab_size.argtypes = [C.c_void_p]
        L.wpo_vocab_size.restype = C.c_int64
        L.wpo_vocab_unk_id.argtypes = [C.c_void_p]
        L.wpo_vocab_token_flags.argtypes = [C.c_void_p, C.c_int64]
        L.wpo_vocab_token_len.argtypes = [C.c_void_p, C.c_int64]
        L.wpo_vocab_token_len.restype = C.c_int64
        L.wpo_vocab_token_word.argtypes = [C.c_void_p, C.c_int64]
        L.wpo_vocab_token_word.restype = C.POINTER(C.c_uint32)
        L.wpo_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_int32)),
                                 C.POINTER(C.c_size_t)]
        L.wpo_encode_mt.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int,
                                    C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_size_t)]
        L.wpo_fast_encode_mt.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int,
                                         C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_size_t)]
        L.wpo_encode_debug.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(_Debug)]
        L.wpo_debug_free.argtypes = [C.POINTER(_Debug)]
        L.wpo_free.argtypes = [C.c_void_p]
        L.wpo_decode_utf8.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.wpo_decode_utf8.restype = C.c_size_t
        L.wpo_suffix_array.argtypes = [C.POINTER(C.c_int32), C.c_int64, C.c_int32, C.POINTER(C.c_int32)]
        L.wpo_kasai.argtypes = [C.POINTER(C.c_int32)] * 3 + [C.c_int64, C.POINTER(C.c_int32)]
        L.wpo_use_libsais.argtypes = [C.c_char_p]
        L.wpo_strerror.argtypes = [C.c_int]
        L.wpo_strerror.restype = C.c_char_p
        for f in ("wpo_is_space", "wpo_is_punctuation", "wpo_is_chinese", "wpo_is_spacing_char"):
            getattr(L, f).argtypes = [C.c_uint32]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _pack(lines):
    lines = [w if isinstance(w, (bytes, bytearray)) else w.encode("utf8") for w in lines]
    off = np.zeros(len(lines) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(w) for w in lines])
    return b"".join(lines), off


class Vocab:
    def __init__(self, lines):
        buf, off = _pack(lines)
        self._h = C.c_void_p()
        rc = lib().wpo_vocab_create(buf, off.ctypes.data_as(C.POINTER(C.c_int64)), len(lines), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise OracleError(lib().wpo_strerror(rc).decode())
        self.size = len(lines)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().wpo_vocab_destroy(self._h)
            self._h = None

    @property
    def unk_id(self):
        return lib().wpo_vocab_unk_id(self._h)

    def flags(self, i):
        return lib().wpo_vocab_token_flags(self._h, i)

    def word(self, i):
        n = lib().wpo_vocab_token_len(self._h, i)
        p = lib().wpo_vocab_token_word(self._h, i)
        return [p[k] for k in range(n)]

    def encode(self, text, threads=1):
        text = text if isinstance(text, (bytes, bytearray)) else text.encode("utf8")
        ids = C.POINTER(C.c_int32)()
        n = C.c_size_t()
        if threads == 1:
            rc = lib().wpo_encode(self._h, text, len(text), C.byref(ids), C.byref(n))
        else:
            rc = lib().wpo_encode_mt(self._h, text, len(text), threads, C.byref(ids), C.byref(n))
        if rc != 0:
            raise OracleError(lib().wpo_strerror(rc).decode())
        out = np.ctypeslib.as_array(ids, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int32)
        lib().wpo_free(ids)
        return out

    def fast_encode(self, text, threads=1):
        """word_piece::fast::encode restated (fast.cpp:19-158)."""
        text = text if isinstance(text, (bytes, bytearray)) else text.encode("utf8")
        ids = C.POINTER(C.c_int32)()
        n = C.c_size_t()
        rc = lib().wpo_fast_encode_mt(self._h, text, len(text), threads, C.byref(ids), C.byref(n))
        if rc != 0:
            raise OracleError(lib().wpo_strerror(rc).decode())
        out = np.ctypeslib.as_array(ids, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int32)
        lib().wpo_free(ids)
        return out

    def encode_debug(self, text):
        """Returns a dict of numpy copies of every intermediate array."""
        text = text if isinstance(text, (bytes, bytearray)) else text.encode("utf8")
        d = _Debug()
        rc = lib().wpo_encode_debug(self._h, text, len(text), C.byref(d))
        if rc != 0:
            raise OracleError(lib().wpo_strerror(rc).decode())
        n = d.n

        def arr(p, k):
            return np.ctypeslib.as_array(p, shape=(k,)).copy() if k > 0 and p else np.zeros(0, np.int32)

        out = {"n_text": d.n_text, "n": n, "longest": d.longest, "alphabet_size": d.alphabet_size}
        for name in ("S", "SA", "rank", "who", "best_left_prefix", "best_right_prefix",
                     "best_left_suffix", "best_right_suffix"):
            out[name] = arr(getattr(d, name), n)
        out["lcp"] = arr(d.lcp, max(n - 1, 0))
        out["ids"] = arr(d.ids, d.n_ids)
        lib().wpo_debug_free(C.byref(d))
        return out


def encode(text, vocab_lines, threads=1):
    return Vocab(vocab_lines).encode(text, threads)


def decode_utf8(b):
    out = np.zeros(len(b) + 1, dtype=np.uint32)
    inv = C.c_int()
    n = lib().wpo_decode_utf8(bytes(b), len(b), out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(inv))
    return out[:n].copy(), bool(inv.value)


def suffix_array(S, alphabet_size=None):
    S = np.ascontiguousarray(S, dtype=np.int32)
    SA = np.zeros(len(S), dtype=np.int32)
    k = int(S.max()) + 1 if alphabet_size is None and len(S) else (alphabet_size or 1)
    rc = lib().wpo_suffix_array(S.ctypes.data_as(C.POINTER(C.c_int32)), len(S), k,
                                SA.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise OracleError("SACA return code: %d" % rc)
    return SA


def use_libsais(enable=True):
    """Route the oracle's SA stage through the reference's libsais (oracle/_ref)."""
    if enable:
        if not os.path.exists(LIBSAIS_REF):
            return False
        return lib().wpo_use_libsais(LIBSAIS_REF.encode()) == 0
    lib().wpo_use_builtin_sa()
    return True
