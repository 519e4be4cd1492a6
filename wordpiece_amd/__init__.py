"""wordpiece_amd — MI355X-native Linear WordPiece (the src/linear.cpp path of gleb-kov/wordpiece).

Host-side mirror of the reference's `word_piece::linear` API on top of the C ABI in
include/wordpiece_amd.h (libwordpiece_amd.so, hand-written HIP for gfx950):

    from wordpiece_amd import linear
    ids = linear.encode("self-made", ["self", "made", "-", "##made"])     # word_piece.hpp:12
    ids = linear.encode("text.txt", "vocab.txt")                           # word_piece.hpp:14
    linear.encodeExternal("text.txt", "vocab.txt", "ids.txt", 500_000_000)  # word_piece.hpp:16

There is no CPU fallback: without the built extension or without a GPU every call raises.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WP_LIB: alternative build of the same library (tuning experiments: profiles/ab.sh)
LIB_PATH = os.environ.get("WP_LIB") or os.path.join(_HERE, "libwordpiece_amd.so")

WP_OPT_FULL_DEPTH, WP_OPT_DEVICE, WP_OPT_KEEP_DEBUG, WP_OPT_STAGE_TIMING, WP_OPT_LCP_KASAI = 1, 2, 3, 4, 5
WP_OPT_COVER_ANCHORS, WP_OPT_ARENA_GUARD, WP_OPT_DEVICES, WP_OPT_VOCAB_IN_S = 7, 8, 9, 10
WP_OPT_SPARSE_EMIT = 11

# every symbol include/wordpiece_amd.h declares (checked by the CPU test-suite)
ABI_SYMBOLS = [
    "wp_vocab_create", "wp_vocab_create_packed", "wp_vocab_from_file", "wp_vocab_destroy", "wp_vocab_size",
    "wp_vocab_unk_id", "wp_vocab_token_flags", "wp_vocab_token_len", "wp_linear_encode",
    "wp_linear_encode_device", "wp_linear_encode_file", "wp_linear_encode_external", "wp_set_option",
    "wp_get_stats", "wp_linear_debug_fetch", "wp_free", "wp_last_error", "wp_device_count",
    "wp_linear_encode_multi", "wp_reserve", "wp_fast_encode", "wp_fast_encode_device", "wp_fast_encode_file",
    "wp_fast_encode_external", "wp_vocab_token_utf8", "wp_trim", "wp_linear_encode_batch", "wp_linear_encode_stream",
]


class WordPieceError(RuntimeError):
    """Mirrors the std::runtime_error the reference throws (message = wp_last_error())."""


class Stats(C.Structure):
    _fields_ = [("n_bytes", C.c_int64), ("n_text", C.c_int64), ("n_total", C.c_int64), ("alphabet", C.c_int64),
                ("longest_token", C.c_int64), ("n_ids", C.c_int64),
                ("symbol_bits", C.c_int32), ("symbols_per_key", C.c_int32), ("rounds", C.c_int32),
                ("sorted_depth", C.c_int32), ("full_depth", C.c_int32),
                ("radix_pass_elems", C.c_int64), ("radix_passes", C.c_int32),
                ("active_per_round", C.c_int64 * 40),
                ("ms_total", C.c_double), ("ms_decode", C.c_double), ("ms_sa", C.c_double), ("ms_lcp", C.c_double),
                ("ms_scan", C.c_double), ("ms_walk", C.c_double), ("ms_radix_scatter", C.c_double),
                ("n_anchors", C.c_int64), ("anchor_mode", C.c_int32), ("ms_h2d", C.c_double), ("ms_d2h", C.c_double),
                ("radix_digit_bytes", C.c_int64), ("ms_host_total", C.c_double), ("guard_zones", C.c_int32),
                ("n_devices", C.c_int32), ("vocab_in_s", C.c_int32), ("reserved0", C.c_int32),
                ("needed_after_round0", C.c_int64), ("key_bits", C.c_int32), ("staged_emit", C.c_int32),
                ("rank_in_pass", C.c_int32), ("trie_refine", C.c_int32), ("arena_bytes", C.c_int64),
                ("list_retries", C.c_int32), ("hist_in_keys", C.c_int32), ("radix_pass_bytes", C.c_int64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "active_per_round"}
        d["active_per_round"] = [int(x) for x in self.active_per_round[:max(self.rounds, 0)]]
        return d


_TEXT_SOURCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
_IDS_SINK = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.c_size_t)

_lib = None


def lib():
    """Loads the HIP extension; fails loudly if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WordPieceError("HIP extension missing: %s (run `python -m wordpiece_amd.build`)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i32p = C.c_void_p, C.POINTER(C.c_int32)
        L.wp_vocab_create.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(vp)]
        L.wp_vocab_create_packed.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.c_int64, C.POINTER(vp)]
        L.wp_vocab_from_file.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.wp_vocab_destroy.argtypes = [vp]
        L.wp_vocab_destroy.restype = None
        L.wp_vocab_size.argtypes = [vp]
        L.wp_vocab_size.restype = C.c_int64
        L.wp_vocab_unk_id.argtypes = [vp]
        L.wp_vocab_unk_id.restype = C.c_int32
        L.wp_vocab_token_flags.argtypes = [vp, C.c_int64]
        L.wp_vocab_token_flags.restype = C.c_int32
        L.wp_vocab_token_len.argtypes = [vp, C.c_int64]
        L.wp_vocab_token_len.restype = C.c_int64
        L.wp_linear_encode.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(i32p), C.POINTER(C.c_size_t)]
        L.wp_linear_encode_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.wp_linear_encode_multi.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.c_int, C.POINTER(i32p),
                                             C.POINTER(C.c_size_t)]
        L.wp_reserve.argtypes = [vp, C.c_size_t]
        L.wp_linear_encode_stream.argtypes = [vp, _TEXT_SOURCE, _IDS_SINK, vp]
        L.wp_linear_encode_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(i32p),
                                             C.POINTER(C.c_size_t)]
        L.wp_trim.argtypes = [vp]
        L.wp_fast_encode.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(i32p), C.POINTER(C.c_size_t)]
        L.wp_fast_encode_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.wp_fast_encode_file.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(i32p), C.POINTER(C.c_size_t)]
        L.wp_fast_encode_external.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.wp_vocab_token_utf8.argtypes = [vp, C.c_int64, C.c_char_p, C.c_size_t]
        L.wp_vocab_token_utf8.restype = C.c_int64
        L.wp_linear_encode_file.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(i32p), C.POINTER(C.c_size_t)]
        L.wp_linear_encode_external.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.wp_set_option.argtypes = [vp, C.c_int, C.c_int64]
        L.wp_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.wp_linear_debug_fetch.argtypes = [vp, C.c_int, i32p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.wp_free.argtypes = [vp]
        L.wp_free.restype = None
        L.wp_last_error.restype = C.c_char_p
        L.wp_device_count.restype = C.c_int
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise WordPieceError(lib().wp_last_error().decode("utf8", "replace"))


def _bytes(x):
    return bytes(x) if isinstance(x, (bytes, bytearray, memoryview)) else x.encode("utf8")


class Vocab:
    """Opaque vocabulary handle (wp_vocab): parsed like utils.cpp:81-137, cached on the device."""

    def __init__(self, lines=None, file=None, device=None):
        self._h = C.c_void_p()
        if file is not None:
            _check(lib().wp_vocab_from_file(_bytes(file), C.byref(self._h)))
        else:
            ls = [_bytes(w) for w in lines]
            off = np.zeros(len(ls) + 1, dtype=np.int64)
            if ls:
                off[1:] = np.cumsum([len(w) for w in ls])
            _check(lib().wp_vocab_create_packed(b"".join(ls), off.ctypes.data_as(C.POINTER(C.c_int64)), len(ls),
                                                C.byref(self._h)))
        if device is not None:
            self.set_option(WP_OPT_DEVICE, device)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.wp_vocab_destroy(self._h)
            self._h = None

    def __len__(self):
        return lib().wp_vocab_size(self._h)

    @property
    def unk_id(self):
        return lib().wp_vocab_unk_id(self._h)

    def token_flags(self, i):
        return lib().wp_vocab_token_flags(self._h, i)

    def token_len(self, i):
        return lib().wp_vocab_token_len(self._h, i)

    def set_option(self, opt, value):
        _check(lib().wp_set_option(self._h, opt, int(value)))

    def stats(self):
        s = Stats()
        _check(lib().wp_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def encode(self, text):
        """Host UTF-8 bytes/str -> numpy int32 ids (wp_linear_encode)."""
        b = _bytes(text)
        ids = C.POINTER(C.c_int32)()
        n = C.c_size_t()
        _check(lib().wp_linear_encode(self._h, b, len(b), C.byref(ids), C.byref(n)))
        return _adopt_ids(ids, n.value)

    def fast_encode(self, text):
        """word_piece::fast::encode on the GPU (wp_fast_encode): host bytes/str -> numpy int32 ids."""
        b = _bytes(text)
        ids = C.POINTER(C.c_int32)()
        n = C.c_size_t()
        _check(lib().wp_fast_encode(self._h, b, len(b), C.byref(ids), C.byref(n)))
        return _adopt_ids(ids, n.value)

    def fast_encode_device(self, d_ptr, nbytes):
        d_ids = C.c_void_p()
        n = C.c_size_t()
        _check(lib().wp_fast_encode_device(self._h, C.c_void_p(d_ptr), nbytes, C.byref(d_ids), C.byref(n)))
        return d_ids.value, n.value

    def token_utf8(self, i):
        """Stored word of vocab line i as UTF-8 bytes (without "##"), or None."""
        n = lib().wp_vocab_token_utf8(self._h, i, None, 0)
        if n < 0:
            return None
        buf = C.create_string_buffer(max(n, 1))
        lib().wp_vocab_token_utf8(self._h, i, buf, n)
        return buf.raw[:n]

    def encode_multi(self, text, devices=None):
        """Host bytes -> numpy int32 ids, sharded over several GPUs behind the C ABI
        (wp_linear_encode_multi).  devices: list of HIP ordinals (may repeat), an int (the first k
        visible GPUs) or None (all visible GPUs)."""
        b = _bytes(text)
        ids = C.POINTER(C.c_int32)()
        n = C.c_size_t()
        if isinstance(devices, (list, tuple)):
            arr = (C.c_int * len(devices))(*devices)
            rc = lib().wp_linear_encode_multi(self._h, b, len(b), arr, len(devices), C.byref(ids), C.byref(n))
        else:
            rc = lib().wp_linear_encode_multi(self._h, b, len(b), None, int(devices or 0), C.byref(ids), C.byref(n))
        _check(rc)
        return _adopt_ids(ids, n.value)

    def encode_batch(self, texts):
        """A sequence of host texts through the shard pipeline (wp_linear_encode_batch): uploads, kernels and id
        downloads of neighbouring texts overlap.  Returns one numpy int32 array per text."""
        bs = [_bytes(t) for t in texts]
        k = len(bs)
        ptrs = (C.c_char_p * k)(*bs)
        sizes = (C.c_size_t * k)(*[len(b) for b in bs])
        ids = (C.POINTER(C.c_int32) * k)()
        ns = (C.c_size_t * k)()
        _check(lib().wp_linear_encode_batch(self._h, ptrs, sizes, k, ids, ns))
        return [_adopt_ids(C.cast(ids[i], C.POINTER(C.c_int32)), ns[i]) for i in range(k)]

    def encode_stream(self, texts, sink):
        """A corpus of any length through the shard pipeline with constant memory (wp_linear_encode_stream): `texts` is
        an iterable of bytes, `sink(index, ids)` receives each text's ids as a numpy view that is valid during the call
        only."""
        it = iter(texts)
        keep = {}

        @_TEXT_SOURCE
        def next_text(_user, index, out_ptr, out_len):
            try:
                b = _bytes(next(it))
            except StopIteration:
                return 0
            keep["text"] = b  # (the one before may go: the library asks for text i + 1 after text i has been uploaded)
            out_ptr[0] = C.cast(C.c_char_p(b), C.c_void_p).value
            out_len[0] = len(b)
            return 1

        @_IDS_SINK
        def got(_user, index, ids, n):
            sink(index, np.ctypeslib.as_array(ids, shape=(n,)) if n else np.zeros(0, dtype=np.int32))

        _check(lib().wp_linear_encode_stream(self._h, next_text, got, None))

    def reserve(self, nbytes):
        """Pre-sizes the device arenas and host staging for inputs of up to nbytes (wp_reserve)."""
        _check(lib().wp_reserve(self._h, int(nbytes)))

    def trim(self):
        """Releases this handle's device arenas, the parked contexts' arenas and the pooled id blocks (wp_trim)."""
        _check(lib().wp_trim(self._h))

    def encode_device(self, d_ptr, nbytes):
        """Text already in HBM at `d_ptr` -> (device pointer of int32 ids, count).  The id buffer is
        owned by the handle and valid until the next call."""
        d_ids = C.c_void_p()
        n = C.c_size_t()
        _check(lib().wp_linear_encode_device(self._h, C.c_void_p(d_ptr), nbytes, C.byref(d_ids), C.byref(n)))
        return d_ids.value, n.value

    def encode_tensor(self, text, copy=True):
        """On-device consumer path: `text` is a uint8 torch tensor on the GPU of this
        handle; returns the ids as an int32 torch tensor on the same device.  copy=False returns a
        zero-copy view of the library's buffer, valid until the next call on this handle.
        (Import torch before the first call into this package: both load a HIP runtime, and torch only
        sees the GPU through its own copy.)"""
        import torch
        if text.dtype != torch.uint8 or not text.is_cuda or not text.is_contiguous():
            raise WordPieceError("encode_tensor needs a contiguous uint8 CUDA/HIP tensor")
        nbytes = text.numel()
        if text.data_ptr() % 4 != 0 or nbytes % 4 != 0:
            # the device entry point reads whole 4-byte words: pad into an aligned staging tensor
            padded = torch.zeros((nbytes + 19) // 16 * 16, dtype=torch.uint8, device=text.device)
            padded[:nbytes] = text
            text = padded
        torch.cuda.current_stream(text.device).synchronize()  # the library runs on its own HIP streams
        d_ids, n = self.encode_device(text.data_ptr(), nbytes)
        if n == 0:
            return torch.zeros(0, dtype=torch.int32, device=text.device)
        view = torch.as_tensor(DeviceIds(d_ids, n), device=text.device)
        return view.clone() if copy else view

    def debug_fetch(self, which, capacity):
        out = np.zeros(max(capacity, 1), dtype=np.int32)
        n = C.c_size_t()
        _check(lib().wp_linear_debug_fetch(self._h, which, out.ctypes.data_as(C.POINTER(C.c_int32)), capacity,
                                           C.byref(n)))
        return out[:n.value]


def _adopt_ids(ids, n):
    """numpy view of the malloc'd id buffer the library returned (no copy); wp_free runs when the
    array (and every view of it) is gone."""
    if n == 0:
        return np.zeros(0, dtype=np.int32)
    out = np.ctypeslib.as_array(ids, shape=(n,))
    weakref.finalize(out.base if out.base is not None else out, lib().wp_free, ids)  # the bottom of the view chain
    return out


class DeviceIds:
    """`__cuda_array_interface__` view of an id buffer in HBM owned by the library (torch.as_tensor,
    cupy.asarray, numba … accept it without a copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}


class _Linear:
    """word_piece::linear of the reference (src/word_piece.hpp:10-21)."""

    @staticmethod
    def encode(text, vocab):
        if isinstance(vocab, (str, bytes)):  # (text_file, vocab_file) overload, linear.cpp:337-341
            ids = C.POINTER(C.c_int32)()
            n = C.c_size_t()
            _check(lib().wp_linear_encode_file(_bytes(text), _bytes(vocab), C.byref(ids), C.byref(n)))
            return _adopt_ids(ids, n.value).tolist()
        return Vocab(vocab).encode(text).tolist()  # linear.cpp:332-335

    @staticmethod
    def encodeExternal(text_file, vocab_file, out_file, memory_limit):  # linear.cpp:343-374
        _check(lib().wp_linear_encode_external(_bytes(text_file), _bytes(vocab_file), _bytes(out_file),
                                               int(memory_limit)))


linear = _Linear()


class _Fast:
    """word_piece::fast of the reference (src/word_piece.hpp:23-36, src/fast.cpp:159-220)."""

    @staticmethod
    def encode(text, vocab):
        if isinstance(vocab, (str, bytes)):  # (text_file, vocab_file) overload, fast.cpp:166-170
            ids = C.POINTER(C.c_int32)()
            n = C.c_size_t()
            _check(lib().wp_fast_encode_file(_bytes(text), _bytes(vocab), C.byref(ids), C.byref(n)))
            return _adopt_ids(ids, n.value).tolist()
        return Vocab(vocab).fast_encode(text).tolist()  # fast.cpp:161-164

    @staticmethod
    def decode(vocab_file, ids):  # fast.cpp:172-187
        import sys
        v = Vocab(file=vocab_file)
        out = []
        for i in ids:
            if i < 0 or i >= len(v):
                print("no token %d" % i, file=sys.stderr)
                continue
            flags = v.token_flags(i)
            if flags & 4:
                print("trying to access malformed token", file=sys.stderr)
                continue
            w = v.token_utf8(i)
            out.append(w if flags & 1 else b"##" + w)
        return out

    @staticmethod
    def encodeExternal(text_file, vocab_file, out_file, memory_limit):  # fast.cpp:189-220
        _check(lib().wp_fast_encode_external(_bytes(text_file), _bytes(vocab_file), _bytes(out_file),
                                             int(memory_limit)))


fast = _Fast()


def shard_bounds(data, world_size):
    """Cuts `data` (bytes) into world_size byte ranges at ASCII whitespace: each cut
    is advanced to the next whitespace byte so no word straddles two shards."""
    n = len(data)
    mv = memoryview(data)
    cuts = [0]
    for r in range(1, world_size):
        p = max(cuts[-1], n * r // world_size)
        while p < n and mv[p] not in (9, 10, 11, 12, 13, 32):
            p += 1
        cuts.append(p)
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world_size)]
