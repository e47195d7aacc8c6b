"""Prototype (big-int) of the table-free variable-base path: Edwards -> Montgomery, x-only ladder on
|a'|, Okeya-Sakurai recovery, back to Edwards, exceptional cases by selects.  Checked against
oracle/bigint_model.py on random and edge inputs before it is written in HIP."""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import bigint_model as M
P, L = M.P, M.L
A = 486662
A24 = 121665
def inv(x): return pow(x % P, P - 2, P)
def sqrt(n):
    r = pow(n, (P + 3) // 8, P)
    if (r * r - n) % P: r = r * M.SQRT_M1 % P
    assert (r * r - n) % P == 0
    return r
C = sqrt((-486664) % P)

def effective(a_bytes):
    """(sign, magnitude) of the integer the reference multiplies by (top digit 9..16 dropped)"""
    a = int.from_bytes(a_bytes, "little")
    b = a + int("0" + "8" * 63, 16)
    e63 = b >> 252
    if e63 <= 8:
        return 0, a
    c63 = e63 - (a >> 252)
    v = (a & ((1 << 252) - 1)) - c63 * (1 << 252)
    return (1, -v) if v < 0 else (0, v)

def ladder(k, u1, nbits=256):
    x2, z2, x3, z3 = 1, 0, u1, 1
    swap = 0
    for i in range(nbits - 1, -1, -1):
        bit = (k >> i) & 1
        swap ^= bit
        if swap: x2, x3, z2, z3 = x3, x2, z3, z2
        swap = bit
        a = (x2 + z2) % P; aa = a * a % P; b = (x2 - z2) % P; bb = b * b % P; e = (aa - bb) % P
        c = (x3 + z3) % P; d = (x3 - z3) % P; da = d * a % P; cb = c * b % P
        x3 = (da + cb) ** 2 % P; z3 = u1 * (da - cb) ** 2 % P
        x2 = aa * bb % P; z2 = e * (aa + A24 * e) % P
    if swap: x2, x3, z2, z3 = x3, x2, z3, z2
    return x2, z2, x3, z3

def mul_via_ladder(a_bytes, pt):
    """pt = affine Edwards (x, y) -> affine Edwards a'*pt, following the planned device data flow"""
    x, y = pt
    sign, k = effective(a_bytes)
    # exceptional inputs: identity and the order-2 point (the only points with u in {inf, 0})
    is_id = (x == 0 and y == 1)
    is_o2 = (x == 0 and y == P - 1)
    d = (1 - y) * x % P
    if d == 0: d = 1                     # keeps the batched inversion alive; result replaced below
    di = inv(d)
    u1 = (1 + y) * x % P * di % P        # (1+y)/(1-y)
    v1 = C * u1 % P * (1 - y) % P * di % P   # c*u/x
    x2, z2, x3, z3 = ladder(k, u1)
    # Okeya-Sakurai: Q = (x2:z2) = kP, Q+P = (x3:z3)
    t1 = u1 * z2 % P; t2 = (x2 + t1) % P; t3 = (x2 - t1) ** 2 % P * x3 % P
    t1 = 2 * A * z2 % P; t2 = (t2 + t1) % P; t4 = (u1 * x2 + z2) % P; t2 = t2 * t4 % P
    t1 = t1 * z2 % P; t2 = (t2 - t1) % P * z3 % P
    Yp = (t2 - t3) % P
    t1 = 2 * v1 * z2 % P * z3 % P
    U, V, W = t1 * x2 % P, Yp, t1 * z2 % P            # projective Montgomery (U:V:W)
    # back to projective Edwards: x = c*u/v, y = (u-1)/(u+1)
    X = C * U % P * (U + W) % P; Y = (U - W) * V % P; Z = V * (U + W) % P
    res_inf = (z2 == 0)                                # kP = infinity -> neutral element
    res_negp = (z3 == 0) and not res_inf               # (k+1)P = infinity -> kP = -P
    res_o2 = (x2 == 0) and (z2 != 0)                   # kP = (0,0) -> Edwards (0,-1)
    if res_negp: X, Y, Z = (-x) % P, y, 1
    if res_o2: X, Y, Z = 0, P - 1, 1
    if res_inf: X, Y, Z = 0, 1, 1
    if is_id: X, Y, Z = 0, 1, 1
    if is_o2: X, Y, Z = (0, P - 1, 1) if (k & 1) else (0, 1, 1)
    zi = inv(Z)
    rx, ry = X * zi % P, Y * zi % P
    if sign: rx = (-rx) % P
    return rx, ry

if __name__ == "__main__":
    rnd = random.Random(1)
    import json
    kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "kats.json")))
    bad = 0; n = 0
    for q in kats["quirk_mul"]:
        if not q["ok"]: continue
        pt = M.decode(bytes.fromhex(q["point"]))
        got = M.encode(mul_via_ladder(bytes.fromhex(q["scalar"]), pt))
        n += 1
        if got.hex() != q["out"]:
            bad += 1; print("MISMATCH", q["point"][:16], q["scalar"][-8:], got.hex()[:16], q["out"][:16])
    for _ in range(300):
        s = bytes(rnd.getrandbits(8) for _ in range(32))
        pt = M.point_mul(bytes(rnd.getrandbits(8) for _ in range(32)))
        if rnd.random() < 0.3:   # mixed-order point
            t8 = M.decode(bytes.fromhex(kats["weak_keys"][rnd.choice([0, 2, 3, 4])]))
            pt = M.add(pt, t8)
        n += 1
        if mul_via_ladder(s, pt) != M.point_mul(s, pt): bad += 1; print("MISMATCH random")
    # scalars around multiples of L for mixed-order points
    for t in [0, 2, 3, 4]:
        t8 = M.decode(bytes.fromhex(kats["weak_keys"][t]))
        pt = M.add(M.point_mul((7).to_bytes(32, "little")), t8)
        for k in [0, 1, 2, L - 1, L, L + 1, 2 * L, 4 * L - 1, 4 * L, 8 * L - 1, 8 * L, 8 * L + 1]:
            s = k.to_bytes(32, "little")
            n += 1
            if mul_via_ladder(s, pt) != M.point_mul(s, pt): bad += 1; print("MISMATCH mixed", t, k)
    print("cases", n, "bad", bad)


# ---- projective base point (the cooperative small-batch kernel): no inversion before the ladder ----
def ladder_proj(k, U1, W1, nbits=256):
    """u(P) = U1 / W1: x3' = W1 (da+cb)^2, z3' = U1 (da-cb)^2 (the extra product rides in an idle row of level 3)"""
    x2, z2, x3, z3 = 1, 0, U1, W1
    swap = 0
    for i in range(nbits - 1, -1, -1):
        bit = (k >> i) & 1
        swap ^= bit
        if swap: x2, x3, z2, z3 = x3, x2, z3, z2
        swap = bit
        a = (x2 + z2) % P; aa = a * a % P; b = (x2 - z2) % P; bb = b * b % P; e = (aa - bb) % P
        c = (x3 + z3) % P; d = (x3 - z3) % P; da = d * a % P; cb = c * b % P
        x3 = W1 * (da + cb) ** 2 % P; z3 = U1 * (da - cb) ** 2 % P
        x2 = aa * bb % P; z2 = e * (aa + A24 * e) % P
    if swap: x2, x3, z2, z3 = x3, x2, z3, z2
    return x2, z2, x3, z3


def mul_via_ladder_proj(a_bytes, P3):
    """P3 = projective Edwards (X : Y : Z) -> affine Edwards a' * P, image (U : V : W) without any inversion:
    u = (Z+Y)/(Z-Y) = U/W, v = c u Z / X = V/W with U = (Z+Y) X, V = c (Z+Y) Z, W = (Z-Y) X"""
    X, Y, Z = P3
    sign, k = effective(a_bytes)
    zmy, zpy = (Z - Y) % P, (Z + Y) % P
    is_id = (X == 0 and zmy == 0)
    is_o2 = (X == 0 and zpy == 0)
    U1, V1, W1 = zpy * X % P, C * zpy % P * Z % P, zmy * X % P
    degenerate = (W1 == 0)
    if degenerate: U1, V1, W1 = 1, 1, 1               # any finite point: the result is replaced below
    x2, z2, x3, z3 = ladder_proj(k, U1, W1)
    # Okeya-Sakurai with u = U1/W1, v = V1/W1, everything scaled by W1^2 (the output is projective)
    T1 = U1 * z2 % P
    Wx2 = W1 * x2 % P
    T2 = (Wx2 + T1) % P                               # W t2
    T3 = (Wx2 - T1) ** 2 % P * x3 % P                 # W^2 t3
    a_ = 2 * A * z2 % P
    T2 = (T2 + W1 * a_) % P                           # W (t2 + 2A z2)
    T4 = (U1 * x2 + W1 * z2) % P                      # W t4
    b_ = a_ * z2 % P
    W2 = W1 * W1 % P
    YP = ((T2 * T4 - W2 * b_) % P * z3 - T3) % P      # W^2 Yp
    TT = 2 * V1 * z2 % P * z3 % P                     # W t1
    Uo, Vo, Wo = W1 * TT % P * x2 % P, YP, W1 * TT % P * z2 % P
    Xe = C * Uo % P * (Uo + Wo) % P; Ye = (Uo - Wo) * Vo % P; Ze = Vo * (Uo + Wo) % P
    res_inf = (z2 == 0)
    res_negp = (z3 == 0) and not res_inf
    res_o2 = (x2 == 0) and (z2 != 0)
    # -P from the image: x = c u / v, y = (u - 1)/(u + 1) with (u, v) = (U1, V1)/W1
    nX, nY, nZ = (-C * U1 % P * (U1 + W1)) % P, (U1 - W1) * V1 % P, V1 * (U1 + W1) % P
    if res_negp: Xe, Ye, Ze = nX, nY, nZ
    if res_o2: Xe, Ye, Ze = 0, P - 1, 1
    if res_inf: Xe, Ye, Ze = 0, 1, 1
    if is_id or (degenerate and not is_o2): Xe, Ye, Ze = 0, 1, 1
    if is_o2: Xe, Ye, Ze = (0, P - 1, 1) if (k & 1) else (0, 1, 1)
    zi = inv(Ze)
    rx, ry = Xe * zi % P, Ye * zi % P
    if sign: rx = (-rx) % P
    return rx, ry


if __name__ == "__main__":
    bad = n = 0
    rnd = random.Random(2)
    def proj(pt):
        z = rnd.randrange(1, P)
        return pt[0] * z % P, pt[1] * z % P, z
    for q in kats["quirk_mul"]:
        if not q["ok"]: continue
        pt = M.decode(bytes.fromhex(q["point"]))
        n += 1
        if M.encode(mul_via_ladder_proj(bytes.fromhex(q["scalar"]), proj(pt))).hex() != q["out"]:
            bad += 1; print("MISMATCH proj", q["point"][:16], q["scalar"][-8:])
    for _ in range(300):
        s = bytes(rnd.getrandbits(8) for _ in range(32))
        pt = M.point_mul(bytes(rnd.getrandbits(8) for _ in range(32)))
        if rnd.random() < 0.3:
            pt = M.add(pt, M.decode(bytes.fromhex(kats["weak_keys"][rnd.choice([0, 2, 3, 4])])))
        n += 1
        if mul_via_ladder_proj(s, proj(pt)) != M.point_mul(s, pt): bad += 1; print("MISMATCH proj random")
    for t in [0, 2, 3, 4]:
        t8 = M.decode(bytes.fromhex(kats["weak_keys"][t]))
        pt = M.add(M.point_mul((7).to_bytes(32, "little")), t8)
        for k in [0, 1, 2, L - 1, L, L + 1, 2 * L, 4 * L - 1, 4 * L, 8 * L - 1, 8 * L, 8 * L + 1]:
            n += 1
            if mul_via_ladder_proj(k.to_bytes(32, "little"), proj(pt)) != M.point_mul(k.to_bytes(32, "little"), pt): bad += 1; print("MISMATCH proj mixed", t, k)
    print("projective-base cases", n, "bad", bad)
