"""Deterministic synthetic workloads: regenerated wherever they are needed, never shipped.

scalar_i = SHA-512("kyber-hip/v1/scalar" || le64(seed) || le64(i)) as a little-endian integer mod L
           (uniform in [0, L): the distribution of Scalar::pick, scalar.rs:167-173)
point_i  = (SHA-512("kyber-hip/v1/point" || le64(seed) || le64(i)) mod L) * B
           (uniform in the prime-order subgroup: the distribution of Point::pick, point.rs:145-154)
"""
import hashlib
import struct

import numpy as np

L = 2**252 + 27742317777372353535851937790883648493


def scalars(n: int, seed: int = 1, tag: bytes = b"scalar") -> np.ndarray:
    out = np.empty((n, 32), dtype=np.uint8)
    pre = b"kyber-hip/v1/" + tag + struct.pack("<Q", seed)
    for i in range(n):
        v = int.from_bytes(hashlib.sha512(pre + struct.pack("<Q", i)).digest(), "little") % L
        out[i] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
    return out


def raw256(n: int, seed: int = 1, tag: bytes = b"raw") -> np.ndarray:
    """unreduced 256-bit strings (exercise scalars >= L and >= 2^255)"""
    out = np.empty((n, 32), dtype=np.uint8)
    pre = b"kyber-hip/v1/" + tag + struct.pack("<Q", seed)
    for i in range(n):
        out[i] = np.frombuffer(hashlib.sha512(pre + struct.pack("<Q", i)).digest()[:32], dtype=np.uint8)
    return out


def messages(n: int, seed: int = 1, length: int = 32):
    pre = b"kyber-hip/v1/msg" + struct.pack("<Q", seed)
    return [hashlib.sha256(pre + struct.pack("<Q", i)).digest()[:length] for i in range(n)]
