"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

Independent big-integer model of the Ed25519 hot path of teleconsys/kyber-rs, used to cross-check
oracle/ed25519_oracle.c (the limb-level restatement) on small cases.  It shares no code or number
representation with the C oracle or with the HIP kernels: affine twisted-Edwards arithmetic on Python
ints, `% p`, `% L`, hashlib SHA-512.

What it models, with the reference lines that define the behaviour:
  * Point::mul(s, None / Some(P))   point.rs:207-224 -> ge.rs:442-486 / 508-568, including the
    top-digit quirk (ge.rs:459 + select at ge.rs:423-434 / 488-500): the scalar is recoded into signed
    radix-16 digits, the top digit e[63] is not recentred, and a value of 9..16 (scalar >= 2^255)
    matches no table entry, i.e. contributes nothing.
  * marshal_binary / unmarshal_binary  point.rs:35-51 -> ge.rs:112-179 (non-canonical y accepted,
    bit 255 = sign of x, x = 0 with sign bit accepted).
  * Scalar::set_bytes / marshal_binary  scalar.rs:175-177, 91-100 (little-endian integer mod L).
  * schnorr::sign with explicit nonce  schnorr_sig.rs:25-47, 128-141; EdDSA keygen/sign
    curve.rs:74-87, eddsa_sig.rs:120-152.
"""
import hashlib

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def inv(x):
    return pow(x, P - 2, P)


def recover_x(y, sign):
    """ge.rs:124-179 — returns x or None.  y is used as given mod p (may be >= p on the wire)."""
    y %= P
    u = (y * y - 1) % P
    v = (D * y * y + 1) % P
    x = (u * pow(v, 3, P) * pow(u * pow(v, 7, P), (P - 5) // 8, P)) % P
    if (v * x * x - u) % P != 0:
        if (v * x * x + u) % P != 0:
            return None
        x = (x * SQRT_M1) % P
    if (x & 1) != sign:
        x = (P - x) % P  # x == 0 stays 0: "x=0 with sign bit set" is accepted
    return x


BY = (4 * inv(5)) % P
BX = recover_x(BY, 0)
B = (BX, BY)
IDENT = (0, 1)


def add(p, q):
    """Affine addition (complete twisted-Edwards law, a = -1)."""
    x1, y1 = p
    x2, y2 = q
    k = D * x1 * x2 * y1 * y2 % P
    x3 = (x1 * y2 + x2 * y1) * inv(1 + k) % P
    y3 = (y1 * y2 + x1 * x2) * inv(1 - k) % P
    return (x3, y3)


def neg(p):
    return ((P - p[0]) % P, p[1])


def _ext_add(p, q):
    # extended coordinates (X:Y:Z:T), unified add-2008-hwcd-3 — inversion-free inner loop
    x1, y1, z1, t1 = p
    x2, y2, z2, t2 = q
    a = (y1 - x1) * (y2 - x2) % P
    b = (y1 + x1) * (y2 + x2) % P
    c = 2 * D * t1 * t2 % P
    d = 2 * z1 * z2 % P
    e, f, g, h = b - a, d - c, d + c, b + a
    return (e * f % P, g * h % P, f * g % P, e * h % P)


def mul_int(k, p):
    """k * p for a (possibly negative) Python int k by double-and-add; returns affine."""
    if k < 0:
        return mul_int(-k, neg(p))
    r = (0, 1, 1, 0)
    q = (p[0], p[1], 1, p[0] * p[1] % P)
    while k:
        if k & 1:
            r = _ext_add(r, q)
        q = _ext_add(q, q)
        k >>= 1
    zi = inv(r[2])
    return (r[0] * zi % P, r[1] * zi % P)


def recode(a_bytes):
    """ge.rs:443-459: 64 signed radix-16 digits; e[63] keeps its carry un-recentred."""
    e = []
    for b in a_bytes:
        e += [b & 15, (b >> 4) & 15]
    carry = 0
    for i in range(63):
        e[i] += carry
        carry = (e[i] + 8) >> 4
        e[i] -= carry << 4
    e[63] += carry
    return e


def effective_scalar(a_bytes):
    """The integer the reference's scalar-mult routines actually multiply by."""
    e = recode(a_bytes)
    top = e[63] if 0 <= e[63] <= 8 else 0  # 9..16 match no table entry
    return sum(e[i] << (4 * i) for i in range(63)) + (top << 252)


def encode(p):
    x, y = p
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


def decode(b):
    """unmarshal_binary: returns (x, y) or None."""
    v = int.from_bytes(b, "little")
    sign = v >> 255
    y = v & ((1 << 255) - 1)
    x = recover_x(y, sign)
    if x is None:
        return None
    return (x, y % P)


def point_mul(scalar_bytes, point=None):
    """Point::mul(s, None) when point is None, else Point::mul(s, Some(point)); returns affine."""
    return mul_int(effective_scalar(scalar_bytes), B if point is None else point)


def sc_from_bytes_mod_l(b):
    return int.from_bytes(b, "little") % L


def sc_bytes(v):
    return (v % L).to_bytes(32, "little")


def schnorr_sign(x_bytes, k_bytes, msg):
    r_enc = encode(point_mul(k_bytes))
    a_enc = encode(point_mul(x_bytes))
    h = int.from_bytes(hashlib.sha512(r_enc + a_enc + msg).digest(), "little") % L
    s = (int.from_bytes(k_bytes, "little") + int.from_bytes(x_bytes, "little") * h) % L
    return r_enc + s.to_bytes(32, "little")


def eddsa_expand(seed):
    d = bytearray(hashlib.sha512(seed).digest())
    d[0] &= 0xF8
    d[31] &= 0x7F
    d[31] |= 0x40
    return bytes(d[:32]), bytes(d[32:])


def eddsa_sign(seed, msg):
    secret, prefix = eddsa_expand(seed)
    r = int.from_bytes(hashlib.sha512(prefix + msg).digest(), "little") % L
    return schnorr_sign(secret, r.to_bytes(32, "little"), msg)


def eddsa_public(seed):
    secret, _ = eddsa_expand(seed)
    return encode(point_mul(secret))


def embed(data, stream):
    """Point::embed / Point::pick (point.rs:90-92, 106-167) over a replayed key stream (bytes, 32 per candidate); data None = pick.
    Returns (affine point, blocks consumed) or (None, -1) when the stream runs out.  Group-theoretic restatement: a candidate that
    decodes is accepted by pick when 8 P != O (and 8 P is returned), by embed when L P == O (and P itself is returned)."""
    dl = 0 if data is None else min(29, len(data))
    for i in range(len(stream) // 32):
        b = bytearray(stream[32 * i:32 * i + 32])
        if data is not None:
            b[0] = dl
            b[1:1 + dl] = data[:dl]
        pt = decode(bytes(b))
        if pt is None:
            continue
        if data is None:
            q = mul_int(8, pt)
            if q == IDENT:
                continue
            return q, i + 1
        if mul_int(L, pt) == IDENT:
            return pt, i + 1
    return None, -1


# ---- verification (eddsa_sig.rs:159-212 / schnorr_sig.rs:53-110); status codes as in ed25519_oracle.c ----
def _order8_ys():
    # y-coordinates of the two order-8 point classes, derived from the group law
    y = 2
    while True:
        pt = decode(y.to_bytes(32, "little"))
        if pt is not None:
            q = mul_int(L, pt)
            if mul_int(4, q) == (0, P - 1):
                return sorted({q[1], (P - q[1]) % P})
        y += 1


WEAK_Y = None


def has_small_order(pt):
    global WEAK_Y
    if WEAK_Y is None:
        WEAK_Y = {0, 1, P - 1, *_order8_ys()}
    return pt[1] in WEAK_Y


def point_is_canonical(b):
    """point.rs:315-337 as the reference evaluates it: its `d` is (0xED - (1 - b0)) >> 8 in wrapping u16 arithmetic (libsodium has
    0xED - 1 - b0), so with bytes 1..30 = 0xff and b31 & 0x7f = 0x7f every b0 >= 0x14 is reported non-canonical:
    y in [2^255 - 236, 2^255 - 1] = the 19 values >= p AND the 217 canonical values p-217 .. p-1."""
    y = int.from_bytes(b, "little") & ((1 << 255) - 1)
    return y < (1 << 255) - 256 + 0x14


def scalar_is_canonical(b):
    return int.from_bytes(b, "little") < L


def verify(flavor, pub, msg, sig):
    if len(sig) != 64:
        return 1
    if flavor == 0:
        if not scalar_is_canonical(sig[32:]):
            return 2
        if not point_is_canonical(sig[:32]):
            return 3
        r = decode(sig[:32])
        if r is None:
            return 4
        if has_small_order(r):
            return 5
        if not point_is_canonical(pub):
            return 6
        a = decode(pub)
        if a is None:
            return 7
        if has_small_order(a):
            return 8
    else:
        r = decode(sig[:32])
        if r is None:
            return 4
        if not point_is_canonical(sig[:32]):
            return 3
        if has_small_order(r):
            return 5
        if not scalar_is_canonical(sig[32:]):
            return 2
        a = decode(pub)
        if a is None:
            return 7
        if not point_is_canonical(pub):
            return 6
        if has_small_order(a):
            return 8
    renc, aenc = (sig[:32], pub) if flavor == 0 else (encode(r), encode(a))
    h = int.from_bytes(hashlib.sha512(renc + aenc + msg).digest(), "little") % L
    lhs = add(r, mul_int(h, a))
    rhs = point_mul(sig[32:])
    return 0 if lhs == rhs else 9
