"""Big-integer / limb-exact prototype of the constant-time field inversion behind fe_invert_gcd (csrc/fe_invert_gcd.h): Bernstein-Yang
"safegcd" divsteps in the 32-bit formulation (20 batches of 30 divsteps, signed 30-bit limbs, 2x2 transition matrices scaled by 2^30) for
p = 2^255 - 19.  Every intermediate is held in the integer width the device code uses (int32 / uint32 / int64) and checked against it.

  python tools/safegcd_proto.py [cases]      compares with pow(x, p - 2, p) on edge values and random x; prints the limb constants"""
import random
import sys

P = (1 << 255) - 19
M30 = (1 << 30) - 1
M32 = (1 << 32) - 1


def s32(v):
    v &= M32
    return v - (1 << 32) if v >> 31 else v


def s64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def to_limbs(x):          # non-negative x < 2^270 -> nine 30-bit limbs
    return [(x >> (30 * i)) & M30 for i in range(9)]


def from_limbs(l):        # signed limbs -> integer
    return sum(v << (30 * i) for i, v in enumerate(l))


MOD = to_limbs(P)
MODINV30 = pow(P, -1, 1 << 30)


def divsteps_30(zeta, f0, g0):
    u, v, q, r = 1, 0, 0, 1
    f, g = f0 & M32, g0 & M32
    for _ in range(30):
        c1 = M32 if zeta < 0 else 0
        c2 = (-(g & 1)) & M32
        x = ((f ^ c1) - c1) & M32
        y = ((u ^ c1) - c1) & M32
        z = ((v ^ c1) - c1) & M32
        g = (g + (x & c2)) & M32
        q = (q + (y & c2)) & M32
        r = (r + (z & c2)) & M32
        c1 &= c2
        zeta = s32((zeta ^ s32(c1)) - 1)
        f = (f + (g & c1)) & M32
        u = (u + (q & c1)) & M32
        v = (v + (r & c1)) & M32
        g >>= 1
        u = (u << 1) & M32
        v = (v << 1) & M32
    return zeta, (s32(u), s32(v), s32(q), s32(r))


def chk64(v):
    assert -(1 << 63) <= v < (1 << 63), "int64 overflow"
    return v


def update_de(d, e, t):
    u, v, q, r = t
    sd = -1 if d[8] < 0 else 0
    se = -1 if e[8] < 0 else 0
    md = (u & sd) + (v & se)
    me = (q & sd) + (r & se)
    cd = chk64(u * d[0] + v * e[0])
    ce = chk64(q * d[0] + r * e[0])
    md -= (MODINV30 * (cd & M32) + md) & M30
    me -= (MODINV30 * (ce & M32) + me) & M30
    assert -(1 << 31) <= md < (1 << 31) and -(1 << 31) <= me < (1 << 31)
    cd = chk64(cd + MOD[0] * md)
    ce = chk64(ce + MOD[0] * me)
    assert cd & M30 == 0 and ce & M30 == 0
    cd >>= 30
    ce >>= 30
    for i in range(1, 9):
        cd = chk64(cd + u * d[i] + v * e[i] + MOD[i] * md)
        ce = chk64(ce + q * d[i] + r * e[i] + MOD[i] * me)
        d[i - 1] = cd & M30
        e[i - 1] = ce & M30
        cd >>= 30
        ce >>= 30
    assert -(1 << 31) <= cd < (1 << 31) and -(1 << 31) <= ce < (1 << 31)
    d[8], e[8] = cd, ce


def update_fg(f, g, t):
    u, v, q, r = t
    cf = chk64(u * f[0] + v * g[0])
    cg = chk64(q * f[0] + r * g[0])
    assert cf & M30 == 0 and cg & M30 == 0
    cf >>= 30
    cg >>= 30
    for i in range(1, 9):
        cf = chk64(cf + u * f[i] + v * g[i])
        cg = chk64(cg + q * f[i] + r * g[i])
        f[i - 1] = cf & M30
        g[i - 1] = cg & M30
        cf >>= 30
        cg >>= 30
    assert -(1 << 31) <= cf < (1 << 31) and -(1 << 31) <= cg < (1 << 31)
    f[8], g[8] = cf, cg


def normalize(r, sign):
    """r in (-2p, p) as signed limbs -> [0, p), negated first when sign < 0"""
    r = list(r)
    add = -1 if r[8] < 0 else 0
    for i in range(9):
        r[i] += MOD[i] & add
    neg = -1 if sign < 0 else 0
    for i in range(9):
        r[i] = (r[i] ^ neg) - neg
    for i in range(8):
        r[i + 1] += r[i] >> 30
        r[i] &= M30
    add = -1 if r[8] < 0 else 0
    for i in range(9):
        r[i] += MOD[i] & add
    for i in range(8):
        r[i + 1] += r[i] >> 30
        r[i] &= M30
    for v in r:
        assert -(1 << 31) <= v < (1 << 31)
    return r


def invert(x):
    assert 0 <= x < P
    d, e = [0] * 9, [1] + [0] * 8
    f, g = list(MOD), to_limbs(x)
    zeta = -1
    for _ in range(20):
        zeta, t = divsteps_30(zeta, f[0] & M32, g[0] & M32)
        update_de(d, e, t)
        update_fg(f, g, t)
    assert all(v == 0 for v in g), "g did not reach 0"
    fv = from_limbs(f)
    assert fv in (1, -1) or x == 0
    out = normalize(d, f[8])
    val = from_limbs(out)
    assert 0 <= val < P
    return val


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    rnd = random.Random(5)
    xs = [0, 1, 2, 3, 19, P - 1, P - 2, P - 19, (P - 1) // 2, (P + 1) // 2, 1 << 254, (1 << 254) - 1, (1 << 255) - 20, 38, 1 << 30, (1 << 30) - 1]
    xs += [1 << k for k in range(0, 255, 7)] + [P - (1 << k) for k in range(1, 254, 11)]
    xs += [rnd.randrange(P) for _ in range(cases)]
    for x in xs:
        want = pow(x, P - 2, P)
        got = invert(x)
        assert got == want, hex(x)
    print(f"{len(xs)} inversions equal to x^(p-2) mod p")
    print("modulus limbs:", [hex(v) for v in MOD])
    print("modulus_inv30:", hex(MODINV30))
