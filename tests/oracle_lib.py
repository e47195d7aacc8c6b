"""ctypes binding of oracle/_build/liboracle.so (TEST INFRASTRUCTURE; see oracle/ed25519_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")


def build():
    src = os.path.join(ORACLE_DIR, "ed25519_oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Oracle:
    def __init__(self):
        self.lib = ctypes.CDLL(build())
        self.lib.orc_init()
        self.lib.orc_decode.restype = ctypes.c_int

    @staticmethod
    def _b(n):
        return ctypes.create_string_buffer(n)

    def mul_base(self, scalar: bytes) -> bytes:
        o = self._b(32)
        self.lib.orc_mul_base(o, None, scalar)
        return o.raw

    def mul_base_ext(self, scalar: bytes) -> np.ndarray:
        e = np.zeros(40, dtype=np.int32)
        self.lib.orc_mul_base(None, _p(e), scalar)
        return e

    def mul(self, scalar: bytes, pt_ext) -> bytes:
        o = self._b(32)
        self.lib.orc_mul(o, None, scalar, _p(_i32(pt_ext)))
        return o.raw

    def mul_ext(self, scalar: bytes, pt_ext) -> np.ndarray:
        e = np.zeros(40, dtype=np.int32)
        self.lib.orc_mul(None, _p(e), scalar, _p(_i32(pt_ext)))
        return e

    def decode(self, enc: bytes):
        e = np.zeros(40, dtype=np.int32)
        ok = self.lib.orc_decode(_p(e), enc)
        return e, int(ok)

    def encode(self, ext) -> bytes:
        o = self._b(32)
        self.lib.orc_encode(o, _p(_i32(ext)))
        return o.raw

    def add(self, a, b, sub=False) -> np.ndarray:
        o = np.zeros(40, dtype=np.int32)
        self.lib.orc_add(_p(o), _p(_i32(a)), _p(_i32(b)), 1 if sub else 0)
        return o

    def neg(self, a) -> np.ndarray:
        o = np.zeros(40, dtype=np.int32)
        self.lib.orc_neg(_p(o), _p(_i32(a)))
        return o

    def base(self) -> np.ndarray:
        o = np.zeros(40, dtype=np.int32)
        self.lib.orc_base(_p(o))
        return o

    def null(self) -> np.ndarray:
        o = np.zeros(40, dtype=np.int32)
        self.lib.orc_null(_p(o))
        return o

    def sc_muladd(self, a, b, c) -> bytes:
        o = self._b(32)
        self.lib.orc_sc_muladd(o, a, b, c)
        return o.raw

    def sc_reduce64(self, x: bytes) -> bytes:
        o = self._b(32)
        self.lib.orc_sc_reduce64(o, x)
        return o.raw

    def sc_reduce32(self, x: bytes) -> bytes:
        o = self._b(32)
        self.lib.orc_sc_reduce32(o, x)
        return o.raw

    def sha512(self, m: bytes) -> bytes:
        o = self._b(64)
        self.lib.orc_sha512(o, m, ctypes.c_size_t(len(m)))
        return o.raw

    def schnorr_sign(self, x, k, msg) -> bytes:
        o = self._b(64)
        self.lib.orc_schnorr_sign(o, x, k, msg, ctypes.c_size_t(len(msg)))
        return o.raw

    def eddsa_expand(self, seed):
        s, p, pub = self._b(32), self._b(32), self._b(32)
        self.lib.orc_eddsa_expand(s, p, pub, seed)
        return s.raw, p.raw, pub.raw

    def eddsa_sign(self, seed, msg) -> bytes:
        o = self._b(64)
        self.lib.orc_eddsa_sign(o, seed, msg, ctypes.c_size_t(len(msg)))
        return o.raw

    def verify(self, flavor: int, pub: bytes, msg: bytes, sig: bytes) -> int:
        self.lib.orc_verify.restype = ctypes.c_int
        return int(self.lib.orc_verify(flavor, pub, msg, ctypes.c_size_t(len(msg)), sig, ctypes.c_size_t(len(sig))))

    def weak_keys(self):
        o = self._b(160)
        self.lib.orc_weak_keys(o)
        return [o.raw[32 * i:32 * i + 32] for i in range(5)]

    def const_bytes(self, which: int) -> bytes:
        o = self._b(32)
        self.lib.orc_const_bytes(o, which)
        return o.raw

    def base_table_bytes(self, i: int, j: int) -> bytes:
        o = self._b(96)
        self.lib.orc_base_table_bytes(o, i, j)
        return o.raw

    def embed(self, data, stream: bytes):
        """Point::embed(data, rand) (data None = Point::pick) over a replayed key stream -> (ext limbs, encoding, blocks consumed);
        (None, None, -1) when the stream ends before a candidate is accepted"""
        e = np.zeros(40, dtype=np.int32)
        o = self._b(32)
        self.lib.orc_embed.restype = ctypes.c_long
        n = self.lib.orc_embed(_p(e), o, data, ctypes.c_long(-1 if data is None else len(data)), stream, ctypes.c_size_t(len(stream) // 32))
        return (e, o.raw, int(n)) if n > 0 else (None, None, -1)

    def point_data(self, ext):
        """Point::data -> bytes, or None for PointError::EmbedDataLength"""
        o = self._b(29)
        self.lib.orc_point_data.restype = ctypes.c_long
        n = self.lib.orc_point_data(o, _p(_i32(ext)))
        return None if n < 0 else o.raw[:n]

    def point_checks(self, enc: bytes) -> int:
        """bit 0 is_canonical(bytes), bit 1 has_small_order of the decoded point, bit 2 decodes (point.rs:286-337)"""
        self.lib.orc_point_checks.restype = ctypes.c_int
        return int(self.lib.orc_point_checks(enc))

    def point_checks_ext(self, ext) -> int:
        self.lib.orc_point_checks_ext.restype = ctypes.c_int
        return int(self.lib.orc_point_checks_ext(_p(_i32(ext))))

    # ---- batches (numpy) ----
    def mul_base_batch(self, scalars, nthreads: int = 1) -> np.ndarray:
        s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        out = np.empty_like(s)
        self.lib.orc_mul_base_batch(_p(out), _p(s), ctypes.c_size_t(s.shape[0]), nthreads)
        return out

    def mul_batch(self, scalars, pts_ext, nthreads: int = 1) -> np.ndarray:
        s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        out = np.empty_like(s)
        p = _i32(pts_ext).reshape(-1, 40)
        self.lib.orc_mul_batch(_p(out), _p(s), _p(p), ctypes.c_size_t(s.shape[0]), nthreads)
        return out

    def mul_enc_batch(self, scalars, pts_enc, nthreads: int = 1):
        """unmarshal_binary of every operand, then mul -> (encodings, ok); an encoding that does not decode gives ok = 0 and the neutral element"""
        s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        e = np.ascontiguousarray(pts_enc, dtype=np.uint8).reshape(-1, 32)
        assert e.shape[0] == s.shape[0]
        out = np.empty_like(s)
        ok = np.empty((s.shape[0],), dtype=np.uint8)
        self.lib.orc_mul_enc_batch(_p(out), _p(ok), _p(s), _p(e), ctypes.c_size_t(s.shape[0]), nthreads)
        return out, ok

    def encode_batch(self, pts_ext, nthreads: int = 1) -> np.ndarray:
        p = _i32(pts_ext).reshape(-1, 40)
        out = np.empty((p.shape[0], 32), dtype=np.uint8)
        self.lib.orc_encode_batch(_p(out), _p(p), ctypes.c_size_t(p.shape[0]), nthreads)
        return out

    def mul_base_ext_batch(self, scalars) -> np.ndarray:
        s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        if s.shape[0] == 0:
            return np.zeros((0, 40), np.int32)
        return np.stack([self.mul_base_ext(s[i].tobytes()) for i in range(s.shape[0])])

    def schnorr_sign_batch(self, x, k, msgs, nthreads: int = 1) -> np.ndarray:
        xs = np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 32)
        ks = np.ascontiguousarray(k, dtype=np.uint8).reshape(-1, 32)
        n = xs.shape[0]
        off = np.zeros(n + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(m) for m in msgs]).astype(np.uint32)
        blob = np.frombuffer(b"".join(msgs) + b"\0", dtype=np.uint8).copy()
        sig = np.empty((n, 64), dtype=np.uint8)
        self.lib.orc_schnorr_sign_batch(_p(sig), _p(xs), _p(ks), _p(blob), _p(off), ctypes.c_size_t(n), nthreads)
        return sig

    def verify_batch(self, flavor, pubs, msgs, sigs, nthreads: int = 1) -> np.ndarray:
        ps = np.ascontiguousarray(pubs, dtype=np.uint8).reshape(-1, 32)
        ss = np.ascontiguousarray(sigs, dtype=np.uint8).reshape(-1, 64)
        n = ps.shape[0]
        off = np.zeros(n + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(m) for m in msgs]).astype(np.uint32)
        blob = np.frombuffer(b"".join(msgs) + b"\0", dtype=np.uint8).copy()
        st = np.empty((n,), dtype=np.uint8)
        self.lib.orc_verify_batch(_p(st), flavor, _p(ps), _p(blob), _p(off), _p(ss), ctypes.c_size_t(n), nthreads)
        return st

    def pubpoly_eval(self, commits_ext, index: int) -> bytes:
        c = _i32(commits_ext).reshape(-1, 40)
        o = self._b(32)
        self.lib.orc_pubpoly_eval(o, _p(c), ctypes.c_size_t(c.shape[0]), ctypes.c_uint32(index))
        return o.raw

    def pripoly_eval(self, coeffs, index: int) -> bytes:
        c = np.ascontiguousarray(coeffs, dtype=np.uint8).reshape(-1, 32)
        o = self._b(32)
        self.lib.orc_pripoly_eval(o, _p(c), ctypes.c_size_t(c.shape[0]), ctypes.c_uint32(index))
        return o.raw

    def lincomb(self, scalars, pts_ext) -> bytes:
        sc = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        c = _i32(pts_ext).reshape(-1, 40)
        assert sc.shape[0] == c.shape[0]
        o = self._b(32)
        self.lib.orc_lincomb(o, _p(sc), _p(c), ctypes.c_size_t(c.shape[0]))
        return o.raw
