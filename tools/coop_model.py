#!/usr/bin/env python3
"""Lane-level model of the cooperative field arithmetic (csrc/coop25519.h) and of the kernels built on it
(csrc/kernels_coop.hip): 64 lanes as numpy arrays, ds_bpermute as an index gather.  The prototype the HIP code was
written against; `python tools/coop_model.py` checks cmul4 / cnorm / cinv against Python integers, one cooperative
mixed addition against the affine group law and one cooperative ladder step against the RFC 7748 formulas."""
import numpy as np

P = 2**255 - 19
BITS = [26, 25] * 5
POS = [sum(BITS[:i]) for i in range(10)]
P2 = [0x7ffffda, 0x3fffffe, 0x7fffffe, 0x3fffffe, 0x7fffffe, 0x3fffffe, 0x7fffffe, 0x3fffffe, 0x7fffffe, 0x3fffffe]
U32 = np.uint64(0xffffffff)


class C:
    pass


def lane_consts():
    c = C()
    lane = np.arange(64)
    c.lane, c.row, c.k = lane, lane >> 4, lane & 15
    c.active = c.k < 10
    k = np.where(c.active, c.k, 0)
    c.bits = np.where(k & 1, 25, 26).astype(np.uint64)
    c.mask = np.where(c.active, (1 << c.bits.astype(np.int64)) - 1, 0).astype(np.uint64)
    c.mask_next = np.where(c.active, np.where(k & 1, 0x3ffffff, 0x1ffffff), 0).astype(np.uint64)
    c.p2 = np.where(c.active, np.array(P2)[k], 0).astype(np.uint64)
    c.c1 = np.where(c.active, np.where(k == 0, 19, 1), 0).astype(np.uint64)
    c.c2 = np.where(c.active, np.where(k < 2, 19, 1), 0).astype(np.uint64)
    base = c.row << 4
    c.prev1 = np.where(c.active, base + (k + 9) % 10, lane)
    c.prev2 = np.where(c.active, base + (k + 8) % 10, lane)
    c.bidx = [base + i for i in range(10)]
    c.ridx, c.mfac = [], []
    for i in range(10):
        j = (k + 10 - i) % 10
        c.ridx.append(np.where(c.active, base + j, lane))
        wrap = i > k
        oo = (i & 1) & (j & 1)
        c.mfac.append(np.where(c.active, np.where(wrap, 19, 1) * np.where(oo, 2, 1), 0).astype(np.uint64))
    return c


def bperm(idx, v):
    return v[idx]


def rowperm_idx(c, p0, p1, p2, p3):
    src = np.choose(c.row, [p0, p1, p2, p3])
    return (src << 4) | c.k


def cnorm(c, v):
    lo = v & c.mask
    cy = v >> c.bits
    return (lo + bperm(c.prev1, cy) * c.c1) & U32


def ccarry(c, s):
    lo = s & c.mask
    t = s >> c.bits
    mid = t & c.mask_next
    hi = s >> np.uint64(51)
    assert (hi < (1 << 13)).all()
    v = lo + bperm(c.prev1, mid) * c.c1 + bperm(c.prev2, hi) * c.c2
    assert (v < (1 << 32)).all()
    return cnorm(c, v)


def cmul4(c, F, G):
    acc = np.zeros(64, dtype=object)
    for i in range(10):
        fb = bperm(c.bidx[i], F)
        gr = G if i == 0 else bperm(c.ridx[i], G)
        fm = fb * c.mfac[i]
        assert (fm < (1 << 32)).all(), "F not tight"
        acc = acc + fm.astype(object) * gr.astype(object)
    assert all(int(a) < (1 << 63) for a in acc)
    return ccarry(c, np.array([int(a) for a in acc], dtype=np.uint64))


def cadd(a, b):
    return (a + b) & U32


def csub(c, a, b):
    assert (c.p2 >= b).all()
    return (a + (c.p2 - b)) & U32


def quad_from_ints(c, vals):
    """four integers -> quad with canonical limbs"""
    q = np.zeros(64, dtype=np.uint64)
    for r, v in enumerate(vals):
        v %= P
        for k in range(10):
            q[16 * r + k] = (v >> POS[k]) & ((1 << BITS[k]) - 1)
    return q


def ints_from_quad(q):
    return [sum(int(q[16 * r + k]) << POS[k] for k in range(10)) % P for r in range(4)]


def cinv(c, z):
    def sqn(f, n):
        for _ in range(n):
            f = cmul4(c, f, f)
        return f
    z2 = sqn(z, 1); t = sqn(z2, 2); z9 = cmul4(c, t, z); z11 = cmul4(c, z9, z2); t = sqn(z11, 1)
    z5 = cmul4(c, t, z9); t = sqn(z5, 5); z10 = cmul4(c, t, z5); t = sqn(z10, 10); z20 = cmul4(c, t, z10)
    t = sqn(z20, 20); t = cmul4(c, t, z20); t = sqn(t, 10); z50 = cmul4(c, t, z10); t = sqn(z50, 50)
    z100 = cmul4(c, t, z50); t = sqn(z100, 100); t = cmul4(c, t, z100); t = sqn(t, 50); t = cmul4(c, t, z50)
    t = sqn(t, 5)
    return cmul4(c, t, z11)


def ladder_step(c, S, UWQ, swap, bit):
    """one step of k_mul_coop's loop (projective base point: U1 in row 0, W1 in row 2 of UWQ) on the state S = (x2, z2, x3,
    z3); the kernel keeps it as SX = (x2, x2, x3, x3), SZ = (z2, z2, z3, z3).  Returns (S', swap')"""
    I = lambda *p: rowperm_idx(c, *p)
    SX, SZ = bperm(I(0, 0, 2, 2), S), bperm(I(1, 1, 3, 3), S)
    SX, SZ, swap = ladder_step_xz(c, SX, SZ, UWQ, swap, bit)
    fx, fz = bperm(I(0, 0, 2, 2), SX), bperm(I(0, 0, 2, 2), SZ)
    return np.where((c.row & 1) == 1, fz, fx), swap


def ladder_step_xz(c, SX, SZ, UWQ, swap, bit):
    rodd, r0, r1, r2, r3 = (c.row & 1) == 1, c.row == 0, c.row == 1, c.row == 2, c.row == 3
    A24Q = np.where(r3 & (c.k == 0), 121665, 0).astype(np.uint64)
    I = lambda *p: rowperm_idx(c, *p)
    x128 = np.where(c.row < 2, 32, 0)          # lane index units here (the HIP code works in bytes: 128)
    swap ^= bit
    ABraw = np.where(rodd, csub(c, SX, SZ), cadd(SX, SZ))
    AB = cnorm(c, ABraw)
    sx = x128 if swap else 0
    L1 = cmul4(c, bperm(I(0, 1, 3, 1) ^ sx, AB), bperm(I(0, 1, 0, 2) ^ sx, ABraw))
    swap = bit
    W, Z = bperm(I(2, 2, 0, 0), L1), bperm(I(3, 3, 1, 1), L1)
    F2raw = np.where(rodd, csub(c, W, Z), np.where(r2, W, cadd(W, Z)))
    G2 = np.where(r3, A24Q, np.where(r2, Z, F2raw))
    F2 = cnorm(c, F2raw)
    L2 = cmul4(c, F2, G2)
    T3 = bperm(I(1, 3, 0, 0), L2)
    E1, A1 = bperm(I(3, 3, 3, 3), F2), bperm(I(0, 0, 0, 0), L1)
    F3 = np.where(r1, cnorm(c, cadd(T3, A1)), T3)
    L3 = cmul4(c, F3, np.where(r1, E1, UWQ))
    SXn = np.where(c.row < 2, bperm(I(2, 2, 2, 2), L2), bperm(I(2, 2, 2, 2), L3))
    SZn = bperm(I(1, 1, 0, 0), L3)
    return SXn, SZn, swap


def madd(c, h, E):
    r0, r1, r2, r3 = c.row == 0, c.row == 1, c.row == 2, c.row == 3
    I = lambda *p: rowperm_idx(c, *p)
    U, V = bperm(I(1, 1, 3, 3), h), bperm(I(0, 0, 0, 0), h)
    FA = cnorm(c, np.where(r0, cadd(U, V), np.where(r1, csub(c, U, V), np.where(r2, U, 0))))
    LA = cmul4(c, FA, E)
    H2 = cadd(h, h)
    Q1 = np.where(c.row < 2, bperm(I(0, 0, 0, 0), LA), bperm(I(2, 2, 2, 2), H2))
    Q2 = bperm(I(1, 1, 2, 2), LA)
    SUM, DIF = cadd(Q1, Q2), csub(c, Q1, Q2)
    FB = cnorm(c, np.where(r0 | r3, bperm(I(0, 0, 0, 0), DIF), bperm(I(2, 2, 2, 2), SUM)))
    GB = np.where(r0 | r2, bperm(I(2, 2, 2, 2), DIF), bperm(I(0, 0, 0, 0), SUM))
    return cmul4(c, FB, GB)


def main():
    import random
    rnd = random.Random(1)
    c = lane_consts()
    for _ in range(50):
        a = [rnd.randrange(P) for _ in range(4)]
        b = [rnd.randrange(P) for _ in range(4)]
        F, G = quad_from_ints(c, a), quad_from_ints(c, b)
        assert ints_from_quad(cmul4(c, F, G)) == [x * y % P for x, y in zip(a, b)]
        G4 = cadd(cadd(G, G), cadd(G, G))                     # lazy second operand, <= 4T
        assert ints_from_quad(cmul4(c, F, G4)) == [4 * x * y % P for x, y in zip(a, b)]
        lazy = csub(c, cadd(F, G), quad_from_ints(c, [1, 2, 3, 4]))
        n = cnorm(c, lazy)
        assert ints_from_quad(n) == [(x + y - d) % P for x, y, d in zip(a, b, [1, 2, 3, 4])]
        assert all(int(n[16 * r + k]) <= (1 << BITS[k]) + 2000 for r in range(4) for k in range(10))
    worst = quad_from_ints(c, [P - 1] * 4)
    assert ints_from_quad(cmul4(c, worst, cadd(cadd(worst, worst), cadd(worst, worst)))) == [4 * (P - 1) * (P - 1) % P] * 4
    a = [rnd.randrange(1, P), 0, 1, P - 1]
    assert ints_from_quad(cinv(c, quad_from_ints(c, a))) == [pow(x, P - 2, P) for x in a]
    # mixed addition against the affine law
    d = -121665 * pow(121666, P - 2, P) % P
    def rand_point():
        while True:
            y = rnd.randrange(P); u = (y * y - 1) % P; v = (d * y * y + 1) % P
            x2 = u * pow(v, P - 2, P) % P
            x = pow(x2, (P + 3) // 8, P)
            if (x * x - x2) % P: x = x * pow(2, (P - 1) // 4, P) % P
            if (x * x - x2) % P == 0: return x, y
    def aff_add(p, q):
        x1, y1 = p; x2, y2 = q
        t = d * x1 * x2 * y1 * y2 % P
        return ((x1 * y2 + x2 * y1) * pow(1 + t, P - 2, P) % P, (y1 * y2 + x1 * x2) * pow(1 - t, P - 2, P) % P)
    for _ in range(5):
        (x1, y1), (x2, y2) = rand_point(), rand_point()
        z = rnd.randrange(1, P)
        h = quad_from_ints(c, [x1 * z, y1 * z, z, x1 * y1 * z])
        E = quad_from_ints(c, [y2 + x2, y2 - x2, 2 * d * x2 * y2, 0])
        X, Y, Z, T = ints_from_quad(madd(c, h, E))
        zi = pow(Z, P - 2, P)
        assert (X * zi % P, Y * zi % P) == aff_add((x1, y1), (x2, y2)) and T * Z % P == X * Y % P
    # ladder step against RFC 7748 with the base point's u = U1 / W1
    for swap0 in (0, 1):
        for bit in (0, 1):
            x2, z2, x3, z3, U1, W1 = (rnd.randrange(P) for _ in range(6))
            S = quad_from_ints(c, [x2, z2, x3, z3])
            UWQ = quad_from_ints(c, [U1, 0, W1, 0])
            S2, sw = ladder_step(c, S, UWQ, swap0, bit)
            s = swap0 ^ bit
            if s: x2, x3, z2, z3 = x3, x2, z3, z2
            A, B, Cc, D = (x2 + z2) % P, (x2 - z2) % P, (x3 + z3) % P, (x3 - z3) % P
            AA, BB, DA, CB = A * A % P, B * B % P, D * A % P, Cc * B % P
            E_ = (AA - BB) % P
            want = [AA * BB % P, E_ * (AA + 121665 * E_) % P, W1 * (DA + CB) ** 2 % P, U1 * (DA - CB) ** 2 % P]
            # the kernel leaves the swap pending: its (x2', z2', x3', z3') are RFC 7748's AFTER that step's own swap
            assert ints_from_quad(S2) == want and sw == bit, (swap0, bit)
    print("coop model: cmul4 / cnorm / cinv / madd / ladder step OK")


if __name__ == "__main__":
    main()
