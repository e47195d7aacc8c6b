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


MONT_C = None   # sqrt(-486664), set by main() from the big-integer prototype


def const_quad(c, vals):
    return quad_from_ints(c, vals)


def prep_quads(c, PQ, cc):
    """k_mul_coop's head: PQ = (X, Y, Z, T) tight -> M = (U, V, W, 0) with U = (Z+Y) X, V = c (Z+Y) Z, W = (Z-Y) X, and the
    quads whose rows the flags are read from (X; Z-Y, Z+Y; W)"""
    I = lambda *p: rowperm_idx(c, *p)
    r0, r1, r2 = c.row == 0, c.row == 1, c.row == 2
    ZZ, YY, XX = bperm(I(2, 2, 2, 2), PQ), bperm(I(1, 1, 1, 1), PQ), bperm(I(0, 0, 0, 0), PQ)
    Fp = cnorm(c, np.where(r0, csub(c, ZZ, YY), cadd(ZZ, YY)))          # (zmy, zpy, zpy, zpy)
    M1 = cmul4(c, Fp, np.where(r2, ZZ, XX))                             # (W, U, t, -)
    M2 = cmul4(c, M1, const_quad(c, [0, 0, cc, 0]))                     # row 2 = V
    a, b = bperm(I(1, 1, 0, 0), M1), bperm(I(2, 2, 2, 2), M2)
    M = np.where(r1, b, np.where(c.row == 3, 0, a))                     # (U, V, W, 0)
    return M, Fp, M1


def recover_quads(c, M, SX, SZ, cc, flags):
    """k_mul_coop's tail (mont_recover_to_edwards_proj in quads): M = (U, V, W, 0), SX = (x2, x2, x3, x3), SZ = (z2, z2, z3, z3)
    after the final swap.  flags = dict(res_inf, res_negp, res_o2, p_id, p_o2, k_odd, negate).  Returns (X, Y, Z, -)."""
    I = lambda *p: rowperm_idx(c, *p)
    r0, r1, r2, r3 = c.row == 0, c.row == 1, c.row == 2, c.row == 3
    one0 = np.where((c.k == 0) & c.active, 1, 0).astype(np.uint64)
    K2A = np.where(r0, one0 * np.uint64(2 * 486662), 0).astype(np.uint64)
    CCQ = const_quad(c, [cc, 0, cc, 0])
    z2a, x2a = bperm(I(0, 0, 0, 0), SZ), bperm(I(0, 0, 0, 0), SX)
    mu, mv, mw = bperm(I(0, 0, 0, 0), M), bperm(I(1, 1, 1, 1), M), bperm(I(2, 2, 2, 2), M)
    up1, um1 = cadd(mu, mw), csub(c, mu, mw)
    # R1: (T1, Wx2, UX, Wz2) = (U z2, W x2, U x2, W z2)
    L1 = cmul4(c, np.where(r0 | r2, mu, mw), np.where(r0 | r3, z2a, x2a))
    # R2: (a_, TT1, t1n, W2) = (2A z2, 2V z2, U c, W^2)
    L2 = cmul4(c, np.where(r0 | r1, z2a, np.where(r2, mu, mw)), np.where(r0, K2A, np.where(r1, cadd(mv, mv), np.where(r2, CCQ, mw))))
    # R3: (T3s, Wa, b_, nXp) = ((Wx2 - T1)^2, W a_, a_ z2, t1n (U + W))
    T3d = cnorm(c, csub(c, bperm(I(1, 1, 1, 1), L1), L1))               # row 0
    a_all = bperm(I(0, 0, 0, 0), L2)
    F3 = np.where(r0, T3d, np.where(r1, mw, np.where(r2, a_all, bperm(I(2, 2, 2, 2), L2))))
    G3 = np.where(r0, T3d, np.where(r1, a_all, np.where(r2, z2a, up1)))
    L3 = cmul4(c, F3, G3)
    # R4: (T3, T2T4, W2b, TT2) = (T3s x3, (Wx2 + T1 + Wa) (UX + Wz2), W2 b_, TT1 z3)
    T2p = cnorm(c, cadd(cadd(L1, bperm(I(0, 0, 0, 0), L1)), L3))        # row 1
    F4 = np.where(r0, L3, np.where(r1, T2p, bperm(I(0, 0, 3, 1), L2)))
    T4 = cadd(bperm(I(2, 2, 2, 2), L1), bperm(I(3, 3, 3, 3), L1))
    G4 = np.where(r0, bperm(I(2, 2, 2, 2), SX), np.where(r1, T4, np.where(r2, L3, SZ)))
    L4 = cmul4(c, F4, G4)
    # R5: (T2z3, t, nY, nZ) = ((T2T4 - W2b) z3, W TT2, V (U - W), V (U + W))
    T2pp = cnorm(c, csub(c, bperm(I(1, 1, 1, 1), L4), bperm(I(2, 2, 2, 2), L4)))
    F5 = np.where(r0, T2pp, np.where(r1, mw, mv))
    G5 = np.where(r0, bperm(I(2, 2, 2, 2), SZ), np.where(r1, bperm(I(3, 3, 3, 3), L4), np.where(r2, um1, up1)))
    L5 = cmul4(c, F5, G5)
    YPn = cnorm(c, csub(c, L5, L4))                                     # row 0: W^2 Yp
    # R6: (Uo, Wo) = (t x2, t z2)
    L6 = cmul4(c, bperm(I(1, 1, 1, 1), L5), np.where(r0, SX, SZ))
    uo, wo = bperm(I(0, 0, 0, 0), L6), bperm(I(1, 1, 1, 1), L6)
    upw, umw = cadd(uo, wo), csub(c, uo, wo)
    # R7: (t1, Y, Z) = (Uo c, (Uo - Wo) V', V' (Uo + Wo))
    vq = bperm(I(0, 0, 0, 0), YPn)
    L7 = cmul4(c, np.where(r0, L6, vq), np.where(r0, CCQ, np.where(r1, umw, upw)))
    # R8: X = t1 (Uo + Wo)
    L8 = cmul4(c, L7, upw)
    RES = np.where(r0, L8, L7)
    # -P = (-t1n (U + W) : (U - W) V : V (U + W))
    NEG = np.where(r0, cnorm(c, csub(c, np.zeros(64, np.uint64), bperm(I(3, 3, 3, 3), L3))), bperm(I(0, 2, 3, 3), L5))
    ID = np.where(r1 | r2, one0, 0).astype(np.uint64)
    O2 = np.where(r1, c.p2 - one0, np.where(r2, one0, 0)).astype(np.uint64)
    RES = np.where(flags["res_negp"], NEG, RES)
    RES = np.where(flags["res_o2"], O2, RES)
    RES = np.where(flags["res_inf"], ID, RES)
    RES = np.where(flags["p_id"], ID, RES)
    RES = np.where(flags["p_o2"], O2 if flags["k_odd"] else ID, RES)
    RES = cnorm(c, RES)
    if flags["negate"]:
        RES = np.where(r0, cnorm(c, csub(c, np.zeros(64, np.uint64), RES)), RES)
    return RES


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
    # head and tail of k_mul_coop in quads against the big-integer prototype (tools/ladder_proto.py)
    import ladder_proto as LP
    def proj_eq(a, b):
        return all((a[i] * b[j] - a[j] * b[i]) % P == 0 for i in range(3) for j in range(i + 1, 3)) and any(v % P for v in a)
    for trial in range(12):
        X, Y, Z = (rnd.randrange(1, P) for _ in range(3))
        M, Fp, M1 = prep_quads(c, quad_from_ints(c, [X, Y, Z, 0]), LP.C)
        U1, V1, W1 = (Z + Y) * X % P, LP.C * (Z + Y) % P * Z % P, (Z - Y) * X % P
        assert ints_from_quad(M)[:3] == [U1, V1, W1]
        assert ints_from_quad(Fp)[:2] == [(Z - Y) % P, (Z + Y) % P] and ints_from_quad(M1)[0] == W1
        k = rnd.getrandbits(255)
        x2, z2, x3, z3 = LP.ladder_proj(k, U1, W1)
        if trial in (3, 4): z3 = 0
        if trial in (5, 6): z2 = 0
        if trial in (7, 8): x2 = 0
        res_inf = z2 == 0; res_negp = z3 == 0 and not res_inf; res_o2 = x2 == 0 and z2 != 0
        for p_id, p_o2, neg in ((0, 0, trial & 1), (1, 0, 0), (0, 1, 1)):
            fl = dict(res_inf=res_inf, res_negp=res_negp, res_o2=res_o2, p_id=p_id, p_o2=p_o2, k_odd=k & 1, negate=neg)
            RES = recover_quads(c, M, quad_from_ints(c, [x2, x2, x3, x3]), quad_from_ints(c, [z2, z2, z3, z3]), LP.C, fl)
            # the prototype's tail on the same values
            T1 = U1 * z2 % P; Wx2 = W1 * x2 % P; T2 = (Wx2 + T1) % P; T3 = (Wx2 - T1) ** 2 % P * x3 % P
            a_ = 2 * 486662 * z2 % P; T2 = (T2 + W1 * a_) % P; T4 = (U1 * x2 + W1 * z2) % P; b_ = a_ * z2 % P; W2 = W1 * W1 % P
            YP = ((T2 * T4 - W2 * b_) % P * z3 - T3) % P; TT = 2 * V1 * z2 % P * z3 % P
            Uo, Vo, Wo = W1 * TT % P * x2 % P, YP, W1 * TT % P * z2 % P
            Xe, Ye, Ze = LP.C * Uo % P * (Uo + Wo) % P, (Uo - Wo) * Vo % P, Vo * (Uo + Wo) % P
            nX, nY, nZ = (-LP.C * U1 % P * (U1 + W1)) % P, (U1 - W1) * V1 % P, V1 * (U1 + W1) % P
            if res_negp: Xe, Ye, Ze = nX, nY, nZ
            if res_o2: Xe, Ye, Ze = 0, P - 1, 1
            if res_inf: Xe, Ye, Ze = 0, 1, 1
            if p_id: Xe, Ye, Ze = 0, 1, 1
            if p_o2: Xe, Ye, Ze = (0, P - 1, 1) if (k & 1) else (0, 1, 1)
            if neg: Xe = (-Xe) % P
            got = ints_from_quad(RES)[:3]
            assert got == [Xe, Ye, Ze], (trial, p_id, p_o2, neg)
    print("coop model: cmul4 / cnorm / cinv / madd / ladder step / prep + recovery in quads OK")


if __name__ == "__main__":
    main()
