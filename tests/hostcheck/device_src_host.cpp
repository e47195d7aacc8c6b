// TEST INFRASTRUCTURE: compiles the DEVICE arithmetic headers (kyber-rs_amd/csrc/*.h) with g++ so
// that tests/test_device_source_on_host.py can run the exact source the HIP kernels inline — limb
// bounds included, every multiply shadowed by a 128-bit overflow check — against the oracle on a
// machine without a GPU.  Built into tests/hostcheck/_build/libdevsrc_host.so by the test itself.
// This library is never loaded by the product (kyber-rs_amd/ loads only libkyber_ed25519_hip.so and
// fails without a GPU); it is not a CPU fallback.
#define KYB_HOST_TEST 1
#include "../../kyber-rs_amd/csrc/scalar_scan.h"
#include <atomic>
#include <cstring>
#include <vector>
#include "../../kyber-rs_amd/csrc/schnorr.h"
#include "../../kyber-rs_amd/csrc/verify.h"
#include "../../kyber-rs_amd/csrc/ge_ladder.h"

static std::atomic<long> g_overflows{0};
extern "C" void kyb_host_overflow(const char*) { g_overflows++; }

using namespace kyb;

static std::vector<uint32_t> g_base_table;
static void ensure_table() {
  if (!g_base_table.empty()) return;
  g_base_table.resize(KYB_BASE_TABLE_WORDS);
  for (int pos = 0; pos < 64; ++pos)
    for (int j = 0; j < 8; ++j) ge_base_table_entry(g_base_table.data(), pos, j);
}
static void load_words(uint32_t w[8], const uint8_t* b) { memcpy(w, b, 32); }

extern "C" {
long hd_overflows() { return g_overflows.load(); }
void hd_sc_pow_mod8L(uint32_t x, uint32_t e, uint8_t mag_out[32], uint32_t* neg_out) {
  uint32_t mag[8], neg;
  sc_pow_mod8L_signed(mag, neg, x, e);
  memcpy(mag_out, mag, 32);
  *neg_out = neg;
}
const uint32_t* hd_base_table() { ensure_table(); return g_base_table.data(); }

void hd_mul_base(uint8_t out[32], const uint8_t scalar[32]) {
  ensure_table();
  uint32_t a[8], w[8];
  load_words(a, scalar);
  tbl_base_words tbl{g_base_table.data()};
  ge_p3 h;
  ge_scalarmult_base(h, a, tbl);
  ge_encode(w, h.X, h.Y, h.Z);
  memcpy(out, w, 32);
}
void hd_mul(uint8_t out[32], int32_t out_ext[40], const uint8_t scalar[32], const int32_t pt[40]) {
  uint32_t a[8], w[8];
  load_words(a, scalar);
  ge_p3 P;
  fe_from_ref10(P.X, pt); fe_from_ref10(P.Y, pt + 10); fe_from_ref10(P.Z, pt + 20); fe_from_ref10(P.T, pt + 30);
  tbl_array_cached tbl;
  ge_p2 r;
  ge_scalarmult(r, a, P, tbl);
  ge_encode(w, r.X, r.Y, r.Z);
  memcpy(out, w, 32);
  if (out_ext) {
    fe zi, x, y, t, one;
    fe_invert(zi, r.Z); fe_mul(x, r.X, zi); fe_mul(y, r.Y, zi); fe_mul(t, x, y); fe_one(one);
    fe_to_ref10(out_ext, x); fe_to_ref10(out_ext + 10, y); fe_to_ref10(out_ext + 20, one); fe_to_ref10(out_ext + 30, t);
  }
}
int hd_decode(int32_t out_ext[40], const uint8_t enc[32]) {
  uint32_t w[8];
  load_words(w, enc);
  ge_p3 h;
  uint32_t ok = ge_decode(h, w);
  fe_to_ref10(out_ext, h.X); fe_to_ref10(out_ext + 10, h.Y); fe_to_ref10(out_ext + 20, h.Z); fe_to_ref10(out_ext + 30, h.T);
  return (int)ok;
}
void hd_encode(uint8_t enc[32], const int32_t pt[40]) {
  ge_p3 P;
  fe_from_ref10(P.X, pt); fe_from_ref10(P.Y, pt + 10); fe_from_ref10(P.Z, pt + 20); fe_from_ref10(P.T, pt + 30);
  uint32_t w[8];
  ge_encode(w, P.X, P.Y, P.Z);
  memcpy(enc, w, 32);
}
void hd_add(int32_t out[40], const int32_t a[40], const int32_t b[40], int sub) {
  ge_p3 A, B, R;
  fe_from_ref10(A.X, a); fe_from_ref10(A.Y, a + 10); fe_from_ref10(A.Z, a + 20); fe_from_ref10(A.T, a + 30);
  fe_from_ref10(B.X, b); fe_from_ref10(B.Y, b + 10); fe_from_ref10(B.Z, b + 20); fe_from_ref10(B.T, b + 30);
  ge_cached c;
  ge_p3_to_cached(c, B);
  ge_cached_cneg(c, sub ? 1u : 0u);
  ge_p1p1 r;
  ge_add(r, A, c);
  ge_p1p1_to_p3(R, r);
  fe_to_ref10(out, R.X); fe_to_ref10(out + 10, R.Y); fe_to_ref10(out + 20, R.Z); fe_to_ref10(out + 30, R.T);
}
void hd_fe_mul(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], int sq) {
  uint32_t wa[8], wb[8], w[8];
  load_words(wa, a); load_words(wb, b);
  fe fa, fb, h;
  fe_from_words(fa, wa); fe_from_words(fb, wb);
  if (sq) fe_sq(h, fa); else fe_mul(h, fa, fb);
  fe_to_words(w, h);
  memcpy(out, w, 32);
}
void hd_fe_invert(uint8_t out[32], const uint8_t a[32]) {
  uint32_t wa[8], w[8];
  load_words(wa, a);
  fe fa, h;
  fe_from_words(fa, wa);
  fe_invert(h, fa);
  fe_to_words(w, h);
  memcpy(out, w, 32);
}
// the divstep inversion (fe_invert_gcd.h) on the same input convention
void hd_fe_invert_gcd(uint8_t out[32], const uint8_t a[32]) {
  uint32_t wa[8], w[8];
  load_words(wa, a);
  fe fa, h;
  fe_from_words(fa, wa);
  fe_invert_gcd(h, fa);
  fe_to_words(w, h);
  memcpy(out, w, 32);
}
// ... and on limbs at the upper end of what the finish kernels hand it (every limb 3.9 x its mask: a value far above p)
void hd_fe_invert_gcd_loose(uint8_t out[32], uint8_t out_fermat[32], const uint8_t a[32]) {
  uint32_t wa[8], w[8];
  load_words(wa, a);
  fe fa, t, h, h2;
  fe_from_words(fa, wa);
  fe_add(t, fa, fa); fe_add(t, t, fa);                // 3 x tight, no carry
  fe_invert_gcd(h, t);
  fe_reduce_weak(t, t);
  fe_invert(h2, t);
  fe_to_words(w, h); memcpy(out, w, 32);
  fe_to_words(w, h2); memcpy(out_fermat, w, 32);
}
// worst-case bound probe: all limbs of f at kf*T, of g at kg*T (in 1/100 units)
void hd_fe_mul_bound_probe(int kf100, int kg100) {
  fe f, g, h;
  for (int i = 0; i < 10; ++i) {
    uint64_t T = (i & 1) ? (1u << 25) : (1u << 26);
    f.v[i] = (uint32_t)(T * kf100 / 100);
    g.v[i] = (uint32_t)(T * kg100 / 100);
  }
  fe_mul(h, f, g);
}
void hd_fe_sq_bound_probe(int kf100) {
  fe f, h;
  for (int i = 0; i < 10; ++i) {
    uint64_t T = (i & 1) ? (1u << 25) : (1u << 26);
    f.v[i] = (uint32_t)(T * kf100 / 100);
  }
  fe_sq(h, f);
}
// the short-fold variants: fe_mul_b6 needs bound(f) * bound(g) <= 6.3, fe_sq_b2 needs f <= 2.5T
void hd_fe_mul_b6_bound_probe(int kf100, int kg100) {
  fe f, g, h;
  for (int i = 0; i < 10; ++i) {
    uint64_t T = (i & 1) ? (1u << 25) : (1u << 26);
    f.v[i] = (uint32_t)(T * kf100 / 100);
    g.v[i] = (uint32_t)(T * kg100 / 100);
  }
  fe_mul_b6(h, f, g);
}
void hd_fe_sq_b2_bound_probe(int kf100) {
  fe f, h;
  for (int i = 0; i < 10; ++i) {
    uint64_t T = (i & 1) ? (1u << 25) : (1u << 26);
    f.v[i] = (uint32_t)(T * kf100 / 100);
  }
  fe_sq_b2(h, f);
}
void hd_sc_muladd(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32]) {
  uint32_t wa[8], wb[8], wc[8], w[8];
  load_words(wa, a); load_words(wb, b); load_words(wc, c);
  sc_muladd(w, wa, wb, wc);
  memcpy(out, w, 32);
}
void hd_sc_reduce512(uint8_t out[32], const uint8_t in[64]) {
  uint32_t x[16], w[8];
  memcpy(x, in, 64);
  sc_reduce512(w, x);
  memcpy(out, w, 32);
}
void hd_sha512(uint8_t out[64], const uint8_t* msg, uint32_t n) {
  sha512_ctx c;
  sha512_init(c);
  sha512_bytes(c, msg, n);
  uint32_t d[16];
  sha512_final(d, c);
  memcpy(out, d, 64);
}
// the same message absorbed in two calls cut anywhere: the second starts at a block position that is not a multiple of 8 (head bytes one by one, then
// eight per step, then the tail)
void hd_sha512_split(uint8_t out[64], const uint8_t* msg, uint32_t n, uint32_t cut) {
  sha512_ctx c;
  sha512_init(c);
  sha512_bytes(c, msg, cut);
  sha512_bytes(c, msg + cut, n - cut);
  uint32_t d[16];
  sha512_final(d, c);
  memcpy(out, d, 64);
}
void hd_schnorr_sign(uint8_t sig[64], const uint8_t x[32], const uint8_t k[32], const uint8_t* msg, uint32_t n) {
  ensure_table();
  uint32_t wx[8], wk[8], s[16];
  load_words(wx, x); load_words(wk, k);
  tbl_base_words tbl{g_base_table.data()};
  schnorr_sign(s, wx, wk, msg, n, tbl);
  memcpy(sig, s, 64);
}
int hd_verify(int flavor, const uint8_t pub[32], const uint8_t* msg, uint32_t n, const uint8_t sig[64]) {
  ensure_table();
  uint32_t wp[8], ws[16], h[8];
  load_words(wp, pub); memcpy(ws, sig, 64);
  ge_p3 R, A;
  uint32_t st = verify_prep(h, R, A, wp, ws, msg, n, flavor);
  // round-trip A through the reference limb layout exactly as the kernels do
  int32_t a_ext[40];
  fe_to_ref10(a_ext, A.X); fe_to_ref10(a_ext + 10, A.Y); fe_to_ref10(a_ext + 20, A.Z); fe_to_ref10(a_ext + 30, A.T);
  ge_p3 A2;
  fe_from_ref10(A2.X, a_ext); fe_from_ref10(A2.Y, a_ext + 10); fe_from_ref10(A2.Z, a_ext + 20); fe_from_ref10(A2.T, a_ext + 30);
  tbl_array_cached tv;
  ge_p2 hA;
  ge_scalarmult(hA, h, A2, tv);
  tbl_base_words tb{g_base_table.data()};
  ge_p3 S;
  ge_scalarmult_base(S, ws + 8, tb);
  ge_p2 sB;
  fe_copy(sB.X, S.X); fe_copy(sB.Y, S.Y); fe_copy(sB.Z, S.Z);
  uint32_t eq = verify_final(R.X, R.Y, hA, sB);
  return (int)((st == 0 && !eq) ? 9u : st);
}
// the A half with the public key given as a point (verify_prep_a_point_with): flags, and whether A is the same group element as the byte form's
int hd_verify_prep_point(const int32_t ext[40], const uint8_t pub[32], const uint8_t* msg, uint32_t n, const uint8_t sig[64], int* same_point, int* on_curve) {
  uint32_t wp[8], ws[16], h1[8], h2[8];
  load_words(wp, pub); memcpy(ws, sig, 64);
  ge_p3 P, A1, A2;
  fe_from_ref10(P.X, ext); fe_from_ref10(P.Y, ext + 10); fe_from_ref10(P.Z, ext + 20); fe_from_ref10(P.T, ext + 30);
  const uint32_t f1 = verify_prep_a_point_with(h1, A1, P, wp, ws, msg, n, ge_decode_fn());
  const uint32_t f2 = verify_prep_a(h2, A2, wp, ws, msg, n);
  *on_curve = (int)ge_on_curve(P);
  *same_point = (int)(ge_equal(A1, A2) && memcmp(h1, h2, 32) == 0);
  return (int)(f1 == f2 ? f1 : 0x100u | f1 | (f2 << 4));
}
void hd_pubpoly_eval(uint8_t out[32], const int32_t* commits, int t, uint32_t index, int nbits) {
  ge_p2 r;
  ge_poly_eval(r, [&](int j, ge_p3& c) {
    const int32_t* p = commits + 40 * j;
    fe_from_ref10(c.X, p); fe_from_ref10(c.Y, p + 10); fe_from_ref10(c.Z, p + 20); fe_from_ref10(c.T, p + 30);
  }, t, index + 1u, nbits);
  uint32_t w[8];
  ge_encode(w, r.X, r.Y, r.Z);
  memcpy(out, w, 32);
}
int hd_equal(const int32_t a[40], const int32_t b[40]) {
  ge_p3 A, B;
  fe_from_ref10(A.X, a); fe_from_ref10(A.Y, a + 10); fe_from_ref10(A.Z, a + 20); fe_from_ref10(A.T, a + 30);
  fe_from_ref10(B.X, b); fe_from_ref10(B.Y, b + 10); fe_from_ref10(B.Z, b + 20); fe_from_ref10(B.T, b + 30);
  return (int)ge_equal(A, B);
}
void hd_mul_ladder_skip(uint8_t out[32], const uint8_t scalar[32], const int32_t pt[40], int skip);
void hd_mul_ladder(uint8_t out[32], const uint8_t scalar[32], const int32_t pt[40]) { hd_mul_ladder_skip(out, scalar, pt, 0); }
void hd_mul_ladder_skip(uint8_t out[32], const uint8_t scalar[32], const int32_t pt[40], int skip) {
  uint32_t a[8], w[8];
  load_words(a, scalar);
  ge_p3 P;
  fe_from_ref10(P.X, pt); fe_from_ref10(P.Y, pt + 10); fe_from_ref10(P.Z, pt + 20); fe_from_ref10(P.T, pt + 30);
  fe d, dinv;
  uint32_t flags;
  mont_prep_den(d, flags, P);
  fe_invert(dinv, d);
  mont_point m;
  mont_prep_finish(m, P, dinv, flags);
  ge_p2 r;
  ge_scalarmult_ladder(r, a, m);
  ge_encode(w, r.X, r.Y, r.Z);
  memcpy(out, w, 32);
}
// the projective-base variant of the same path (no inversion before the ladder): what the cooperative small-batch kernel computes
void hd_mul_ladder_proj(uint8_t out[32], const uint8_t scalar[32], const int32_t pt[40], int skip) {
  uint32_t a[8], w[8];
  load_words(a, scalar);
  ge_p3 P;
  fe_from_ref10(P.X, pt); fe_from_ref10(P.Y, pt + 10); fe_from_ref10(P.Z, pt + 20); fe_from_ref10(P.T, pt + 30);
  ge_p2 r;
  ge_scalarmult_ladder_proj(r, a, P, skip);
  ge_encode(w, r.X, r.Y, r.Z);
  memcpy(out, w, 32);
}
// The four-lane ladder (ge_ladder_quad.h is device code: DPP lane moves) as a HOST MODEL of the same data flow — the four lanes of an item are the
// four elements of an array, quad_perm is indexing, the per-lane selections are those of mont_ladder_quad line by line — so that the formulas and
// the limb BOUNDS of every operation of that kernel run under the overflow-checked host build (kyb_add32 / kyb_sub32 / the column checks of fe_mul).
void hd_mul_ladder_quad_model(uint8_t out[32], const uint8_t scalar[32], const int32_t pt[40], int skip) {
  uint32_t a[8], w[8], neg, mag[8];
  load_words(a, scalar);
  ge_p3 P;
  fe_from_ref10(P.X, pt); fe_from_ref10(P.Y, pt + 10); fe_from_ref10(P.Z, pt + 20); fe_from_ref10(P.T, pt + 30);
  sc_effective(neg, mag, a);
  mont_point_proj m;
  mont_prep_proj(m, P);
  fe one, zero, st[4], U[4];
  uint32_t U19[4][10];
  fe_one(one); fe_zero(zero);
  fe_copy(st[0], m.U); fe_copy(st[1], m.W); fe_copy(st[2], one); fe_copy(st[3], zero);      // x3 | z3 | x2 | z2
  fe_copy(U[0], m.W); fe_copy(U[1], m.U); fe_copy(U[2], one); fe_copy(U[3], one);
  for (int q = 0; q < 4; ++q) fe_x19(U19[q], U[q]);
  uint32_t swap = 0;
  for (int i = 255 - skip; i >= 0; --i) {
    const uint32_t bit = (mag[i >> 5] >> (i & 31)) & 1u;
    swap ^= bit;
    fe sd[4], X[4], F[4], G[4], r1[4], Y1[4], Y2[4], T[4], g[4], r2[4];
    for (int q = 0; q < 4; ++q) { const fe& p = st[q ^ 1]; if (q & 1) fe_sub(sd[q], p, st[q]); else fe_add(sd[q], p, st[q]); }      // c | d | a | b
    const int cross[4] = {2, 3, 1, 0};
    for (int q = 0; q < 4; ++q) fe_copy(X[q], sd[cross[q]]);
    for (int q = 0; q < 4; ++q) {
      const uint32_t hi = (uint32_t)(q >> 1);
      fe_select(F[q], X[q], sd[q], hi | swap);
      fe_select(G[q], F[q], X[q], hi);
    }
    swap = bit;
    for (int q = 0; q < 4; ++q) fe_mul(r1[q], F[q], G[q]);      // aa | bb | da | cb
    for (int q = 0; q < 4; ++q) { fe_copy(Y1[q], r1[q < 2 ? 2 : 0]); fe_copy(Y2[q], r1[q < 2 ? 3 : 1]); }
    for (int q = 0; q < 4; ++q) { if (q & 1) fe_sub(T[q], Y1[q], Y2[q]); else fe_add(T[q], Y1[q], Y2[q]); }
    for (int q = 0; q < 4; ++q) { fe_mul_small(g[q], T[q], q == 3 ? 121665u : 0u); fe_addw(g[q], g[q], Y1[q]); }
    for (int q = 0; q < 4; ++q) {
      fe_select(F[q], T[q], Y1[q], (uint32_t)(q == 2));
      fe_select(G[q], T[q], g[q], (uint32_t)(q == 3));
      fe_select(G[q], G[q], Y2[q], (uint32_t)(q == 2));
    }
    for (int q = 0; q < 4; ++q) fe_mul(r2[q], F[q], G[q]);
    for (int q = 0; q < 4; ++q) fe_mul_g19<true>(st[q], r2[q], U[q], U19[q]);
  }
  if (swap) { fe t; fe_copy(t, st[0]); fe_copy(st[0], st[2]); fe_copy(st[2], t); fe_copy(t, st[1]); fe_copy(st[1], st[3]); fe_copy(st[3], t); }
  ge_p2 r;
  mont_recover_to_edwards_proj(r, m, st[2], st[3], st[0], st[1], mag[0] & 1u, neg);
  ge_encode(w, r.X, r.Y, r.Z);
  memcpy(out, w, 32);
}
// k_finish_wave (kernels_coop.hip) as a HOST MODEL: the 64 lanes of a wavefront are 64 array elements — the butterfly product of the Z's (lane l takes the
// sub-product of lane l ^ 2^j at level j and keeps it), ONE inversion of what every lane then holds, the six kept sub-products multiplied back on;
// a zero Z counts as 1 and gets 0 as its inverse.  enc: 64 x 32 bytes out, pts: 64 x 40 reference limbs in.
void hd_finish_wave_model(uint8_t* enc, const int32_t* pts) {
  fe X[64], Y[64], P[64], Q[6][64], I[64], one, zero;
  uint32_t z_zero[64];
  fe_one(one); fe_zero(zero);
  for (int l = 0; l < 64; ++l) {
    fe Z;
    fe_from_ref10(X[l], pts + 40 * l); fe_from_ref10(Y[l], pts + 40 * l + 10); fe_from_ref10(Z, pts + 40 * l + 20);
    z_zero[l] = 1u - fe_is_nonzero(Z);
    fe_copy(P[l], Z);
    fe_cmov(P[l], one, z_zero[l]);
  }
  for (int j = 0; j < 6; ++j) {
    for (int l = 0; l < 64; ++l) fe_copy(Q[j][l], P[l ^ (1 << j)]);
    for (int l = 0; l < 64; ++l) fe_mul(P[l], P[l], Q[j][l]);
  }
  for (int l = 0; l < 64; ++l) {
    fe_inv(I[l], P[l]);                                  // (the kernel inverts once, from the canonical words every lane agrees on)
    for (int j = 5; j >= 0; --j) fe_mul(I[l], I[l], Q[j][l]);
    fe_cmov(I[l], zero, z_zero[l]);
    fe x, y;
    fe_mul(x, X[l], I[l]);
    fe_mul(y, Y[l], I[l]);
    uint32_t w[8];
    fe_to_words(w, y);
    w[7] ^= fe_is_negative(x) << 31;
    memcpy(enc + 32 * l, w, 32);
  }
}
// the flow of kyb_lincomb_batch for one group: t ladder multiplications, the halving passes of k_pair_sum, encode
void hd_lincomb(uint8_t out[32], const uint8_t* scalars, const int32_t* pts, int t) {
  std::vector<ge_p2> v((size_t)t);
  for (int i = 0; i < t; ++i) {
    uint32_t a[8];
    load_words(a, scalars + 32 * i);
    const int32_t* pt = pts + 40 * i;
    ge_p3 P;
    fe_from_ref10(P.X, pt); fe_from_ref10(P.Y, pt + 10); fe_from_ref10(P.Z, pt + 20); fe_from_ref10(P.T, pt + 30);
    fe d, dinv;
    uint32_t flags;
    mont_prep_den(d, flags, P);
    fe_invert(dinv, d);
    mont_point m;
    mont_prep_finish(m, P, dinv, flags);
    ge_scalarmult_ladder(v[i], a, m);
  }
  for (size_t len = (size_t)t; len > 1;) {
    const size_t half = (len + 1) / 2;
    for (size_t j = 0; j < len - half; ++j) { ge_p2 r; ge_p2_add(r, v[j], v[j + half]); v[j] = r; }
    len = half;
  }
  uint32_t w[8];
  ge_encode(w, v[0].X, v[0].Y, v[0].Z);
  memcpy(out, w, 32);
}
// the flow of k_poly_eval_part + ladder + k_pair_sum for ONE evaluation: the Horner chain cut into segments of `len` coefficients,
// every partial result (negated when the multiplier's representative is) multiplied by |x^(s len) mod 8L| on the ladder, summed, encoded
void hd_pubpoly_eval_segments(uint8_t out[32], const int32_t* commits, int t, uint32_t index, int nbits, int len) {
  const uint32_t x = index + 1u;
  const int segs = (t + len - 1) / len;
  std::vector<ge_p2> v((size_t)segs);
  for (int sg = 0; sg < segs; ++sg) {
    const int lo = sg * len, cnt = t - lo < len ? t - lo : len;
    ge_p3 q;
    ge_poly_eval_p3(q, [&](int j, ge_p3& c) {
      const int32_t* p = commits + 40 * (lo + j);
      fe_from_ref10(c.X, p); fe_from_ref10(c.Y, p + 10); fe_from_ref10(c.Z, p + 20); fe_from_ref10(c.T, p + 30);
    }, cnt, x, nbits);
    uint32_t mag[8], neg;
    sc_pow_mod8L_signed(mag, neg, x, (uint32_t)lo);
    fe nx, nt;
    fe_reduce_weak(q.X, q.X); fe_reduce_weak(q.T, q.T);
    fe_neg(nx, q.X); fe_neg(nt, q.T);
    fe_reduce_weak(nx, nx); fe_reduce_weak(nt, nt);
    fe_cmov(q.X, nx, neg); fe_cmov(q.T, nt, neg);
    int32_t ext[40];                                   // through the reference-limb record, as the kernel hands it to k_mont_prep
    fe_to_ref10(ext, q.X); fe_to_ref10(ext + 10, q.Y); fe_to_ref10(ext + 20, q.Z); fe_to_ref10(ext + 30, q.T);
    ge_p3 P;
    fe_from_ref10(P.X, ext); fe_from_ref10(P.Y, ext + 10); fe_from_ref10(P.Z, ext + 20); fe_from_ref10(P.T, ext + 30);
    fe d, dinv;
    uint32_t flags;
    mont_prep_den(d, flags, P);
    fe_invert(dinv, d);
    mont_point m;
    mont_prep_finish(m, P, dinv, flags);
    ge_scalarmult_ladder(v[sg], mag, m, 1);
  }
  for (size_t n = (size_t)segs; n > 1;) {
    const size_t half = (n + 1) / 2;
    for (size_t j = 0; j < n - half; ++j) { ge_p2 r; ge_p2_add(r, v[j], v[j + half]); v[j] = r; }
    n = half;
  }
  uint32_t w[8];
  ge_encode(w, v[0].X, v[0].Y, v[0].Z);
  memcpy(out, w, 32);
}
int hd_common_leading_zero_bits(const uint8_t* scalars, uint32_t n) { return kyb::common_leading_zero_bits(scalars, n); }
int hd_common_leading_zero_bits_exact(const uint8_t* scalars, uint32_t n) { return kyb::common_leading_zero_bits(scalars, n, 256); }
void hd_effective(uint8_t mag[32], int* neg, const uint8_t scalar[32]) {
  uint32_t a[8], m[8], n;
  load_words(a, scalar);
  sc_effective(n, m, a);
  memcpy(mag, m, 32);
  *neg = (int)n;
}
static std::vector<uint32_t> g_base32_table;
void hd_mul_base32(uint8_t out[32], const uint8_t scalar[32]) {
  if (g_base32_table.empty()) {
    g_base32_table.resize(KYB_BASE32_TABLE_WORDS);
    for (int pos = 0; pos < KYB_BASE32_POS; ++pos)
      for (int j = 0; j < 16; ++j) ge_base32_table_entry(g_base32_table.data(), pos, j);
  }
  uint32_t a[8], w[8];
  load_words(a, scalar);
  tbl_base32_words tbl{g_base32_table.data()};
  ge_p3 h;
  ge_scalarmult_base32(h, a, tbl);
  ge_encode(w, h.X, h.Y, h.Z);
  memcpy(out, w, 32);
}
static std::vector<uint32_t> g_base64_table;
void hd_mul_base64(uint8_t out[32], const uint8_t scalar[32]) {
  if (g_base64_table.empty()) {
    g_base64_table.resize(KYB_BASE64_TABLE_WORDS);
    for (int pos = 0; pos < KYB_BASE64_POS; ++pos)
      for (int j = 0; j < (pos == KYB_BASE64_POS - 1 ? 16 : 32); ++j) ge_base64_table_entry(g_base64_table.data(), pos, j);
  }
  uint32_t a[8], w[8];
  load_words(a, scalar);
  tbl_base64_words tbl{g_base64_table.data()};
  ge_p3 h;
  ge_scalarmult_base64(h, a, tbl);
  ge_encode(w, h.X, h.Y, h.Z);
  memcpy(out, w, 32);
}
// the same as k_mul_base64_quarters forms it (kernels_base.hip): four partial sums over a quarter of the windows each, added up as that kernel adds them
// (a staged part comes back as X, Y, Z only)
void hd_mul_base64_quarters(uint8_t out[32], const uint8_t scalar[32]) {
  uint8_t whole[32];
  hd_mul_base64(whole, scalar);          // (builds the table)
  uint32_t a[8], w[8];
  load_words(a, scalar);
  tbl_base64_words tbl{g_base64_table.data()};
  ge_p3 part[4];
  for (int q = 0; q < 4; ++q) ge_scalarmult_base64_part(part[q], a, tbl, 11 * q, q == 3 ? KYB_BASE64_POS : 11 * q + 11);
  auto add_staged = [](ge_p3& h, const ge_p3& other) {
    ge_p2 b;
    fe_copy(b.X, other.X); fe_copy(b.Y, other.Y); fe_copy(b.Z, other.Z);
    ge_p3 B;
    ge_p2_to_p3(B, b);
    ge_cached c;
    ge_p3_to_cached(c, B);
    ge_p1p1 t;
    ge_add(t, h, c);
    ge_p1p1_to_p3(h, t);
  };
  add_staged(part[0], part[2]);
  add_staged(part[1], part[3]);
  add_staged(part[0], part[1]);
  ge_encode(w, part[0].X, part[0].Y, part[0].Z);
  memcpy(out, w, 32);
}
void hd_eddsa_sign(uint8_t sig[64], const uint8_t seed[32], const uint8_t* msg, uint32_t n) {
  ensure_table();
  uint32_t ws[8], x[8], r[8], s[16];
  load_words(ws, seed);
  eddsa_expand_and_nonce(x, r, ws, msg, n);
  tbl_base_words tbl{g_base_table.data()};
  schnorr_sign(s, x, r, msg, n, tbl);
  memcpy(sig, s, 64);
}
void hd_recode(int8_t e[64], const uint8_t scalar[32]) {
  uint32_t a[8];
  load_words(a, scalar);
  sc_digits d;
  sc_recode(d, a);
  for (int i = 0; i < 63; ++i) {
    uint32_t mag, neg;
    sc_digit(mag, neg, d, i);
    e[i] = (int8_t)(neg ? -(int)mag : (int)mag);
  }
  e[63] = (int8_t)d.top;
}
}
