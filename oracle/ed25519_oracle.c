/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product path (kyber-rs_amd/, include/); only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py use it, and only as the checker / the reported CPU baseline.
 *
 * CPU restatement, in plain C, of the Ed25519 hot path of teleconsys/kyber-rs (reference mounted at
 * /root/reference, crate kyber-rs 0.1.0-alpha.9; it is Rust and cannot be compiled in this image —
 * no cargo/rustc — so there is no oracle/_ref build).  Same number representation
 * (ten signed 32-bit limbs, radix 2^25.5, fe.rs:4-8), same formulas and same operation order as
 *   src/group/edwards25519/fe.rs   (field)            src/group/edwards25519/ge.rs (group, scalar mult)
 *   src/group/edwards25519/point.rs (Point surface)   src/group/edwards25519/scalar.rs (mod-L arithmetic)
 *   src/sign/schnorr/schnorr_sig.rs, src/sign/eddsa/eddsa_sig.rs, src/group/edwards25519/curve.rs
 * Each function cites the lines it follows.  Products are written as loops over the limb indices
 * instead of the reference's 100 named terms; the sums, the *19 / *2 placement and the carry order
 * are the reference's, so every intermediate limb vector is the one the reference computes.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against every known-answer vector the
 * reference's own tests hold for the path (the 1024-line sign.input.gz golden file,
 * RFC 8032 §7.1, scalar KATs, decode KATs, WEAK_KEYS) and against oracle/bigint_model.py.
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>

typedef int32_t fe[10];
typedef struct { fe X, Y, Z; } ge_p2;
typedef struct { fe X, Y, Z, T; } ge_p3;
typedef struct { fe X, Y, Z, T; } ge_p1p1;
typedef struct { fe ypx, ymx, xy2d; } ge_precomp;
typedef struct { fe YpX, YmX, Z, T2d; } ge_cached;

/* ------------------------------------------------------------------ field: fe.rs ---------- */
static void fe_0(fe h) { memset(h, 0, sizeof(fe)); }                        /* fe.rs:10-14 */
static void fe_1(fe h) { memset(h, 0, sizeof(fe)); h[0] = 1; }              /* fe.rs:16-19 */
static void fe_copy(fe h, const fe f) { memmove(h, f, sizeof(fe)); }
static void fe_add(fe h, const fe f, const fe g) { for (int i = 0; i < 10; i++) h[i] = f[i] + g[i]; }  /* fe.rs:21-30 */
static void fe_sub(fe h, const fe f, const fe g) { for (int i = 0; i < 10; i++) h[i] = f[i] - g[i]; }  /* fe.rs:32-41 */
static void fe_neg(fe h, const fe f) { for (int i = 0; i < 10; i++) h[i] = -f[i]; }                    /* fe.rs:266-270 */
static void fe_cmov(fe f, const fe g, int32_t b) {                           /* fe.rs:43-50 */
  int32_t m = -b;
  for (int i = 0; i < 10; i++) f[i] ^= m & (f[i] ^ g[i]);
}
static int64_t load3(const uint8_t* s) { return (int64_t)s[0] | ((int64_t)s[1] << 8) | ((int64_t)s[2] << 16); }
static int64_t load4(const uint8_t* s) { return load3(s) | ((int64_t)s[3] << 24); }

#define BITS(i) (((i) & 1) ? 25 : 26)
/* rounding carry from limb i into limb i+1 (limb 9 wraps into limb 0 times 19) */
static inline __attribute__((always_inline)) void carry_at(int64_t h[10], int i) {
  int64_t c = (h[i] + ((int64_t)1 << (BITS(i) - 1))) >> BITS(i);
  if (i == 9) h[0] += c * 19; else h[i + 1] += c;
  h[i] -= c * ((int64_t)1 << BITS(i));
}
/* the 12-carry schedule shared by fe_mul / fe_square / fe_square2 (fe.rs:463-523) */
static inline __attribute__((always_inline)) void carry_mul_schedule(fe out, int64_t h[10]) {
  carry_at(h, 0); carry_at(h, 4); carry_at(h, 1); carry_at(h, 5); carry_at(h, 2); carry_at(h, 6);
  carry_at(h, 3); carry_at(h, 7); carry_at(h, 4); carry_at(h, 8); carry_at(h, 9); carry_at(h, 0);
  for (int i = 0; i < 10; i++) out[i] = (int32_t)h[i];
}
/* fe.rs:67-122 — ignores bit 255, accepts values >= p */
static void fe_frombytes(fe dst, const uint8_t* s) {
  int64_t h[10];
  h[0] = load4(s);
  h[1] = load3(s + 4) << 6;
  h[2] = load3(s + 7) << 5;
  h[3] = load3(s + 10) << 3;
  h[4] = load3(s + 13) << 2;
  h[5] = load4(s + 16);
  h[6] = load3(s + 20) << 7;
  h[7] = load3(s + 23) << 5;
  h[8] = load3(s + 26) << 4;
  h[9] = (load3(s + 29) & 8388607) << 2;
  static const int order[10] = {9, 1, 3, 5, 7, 0, 2, 4, 6, 8};
  for (int k = 0; k < 10; k++) carry_at(h, order[k]);
  for (int i = 0; i < 10; i++) dst[i] = (int32_t)h[i];
}
/* fe.rs:147-238 */
static void fe_tobytes(uint8_t* s, const fe f) {
  int32_t h[10];
  memcpy(h, f, sizeof(h));
  int32_t q = (19 * h[9] + (1 << 24)) >> 25;
  for (int i = 0; i < 10; i++) q = (h[i] + q) >> BITS(i);
  h[0] += 19 * q;
  for (int i = 0; i < 10; i++) {
    int32_t c = h[i] >> BITS(i);
    if (i < 9) h[i + 1] += c;
    h[i] -= c * (1 << BITS(i));
  }
  /* pack: limb i starts at bit ceil(25.5 i) */
  uint8_t out[33];
  memset(out, 0, sizeof(out));
  int off = 0;
  for (int i = 0; i < 10; i++) {
    uint64_t v = (uint64_t)(uint32_t)h[i] << (off & 7);
    for (int b = 0; b < 5; b++) out[(off >> 3) + b] |= (uint8_t)(v >> (8 * b));
    off += BITS(i);
  }
  memcpy(s, out, 32);
}
static int fe_isnegative(const fe f) { uint8_t s[32]; fe_tobytes(s, f); return s[0] & 1; }   /* fe.rs:240-245 */
static int fe_isnonzero(const fe f) {                                                       /* fe.rs:247-257 */
  uint8_t s[32]; fe_tobytes(s, f);
  uint8_t x = 0; for (int i = 0; i < 32; i++) x |= s[i];
  return x != 0;
}
/* fe.rs:299-535: h_k = sum_{i+j=k} f_i g_j * (2 if i,j odd) + 19 * sum_{i+j=k+10} (same).
 * The *19 and *2 are 32-bit precomputations exactly as in the reference (g1_19.., f1_2..,
 * fe.rs:320-333): for even k an odd i always meets an odd j, so column k uses the "odd limbs doubled"
 * copy of f; for odd k it uses f itself. */
static void fe_mul(fe h, const fe f, const fe g) {
  int32_t g19[10], fo[10];
  int64_t acc[10];
  for (int i = 0; i < 10; i++) { g19[i] = 19 * g[i]; fo[i] = (i & 1) ? 2 * f[i] : f[i]; }
#pragma GCC unroll 10
  for (int k = 0; k < 10; k++) {
    const int32_t* ff = (k & 1) ? f : fo;
    int64_t s = 0;
#pragma GCC unroll 10
    for (int i = 0; i <= k; i++) s += (int64_t)ff[i] * (int64_t)g[k - i];
#pragma GCC unroll 10
    for (int i = k + 1; i < 10; i++) s += (int64_t)ff[i] * (int64_t)g19[k - i + 10];
    acc[k] = s;
  }
  carry_mul_schedule(h, acc);
}
/* fe.rs:544-688 (and fe.rs:700-855 with dbl=1: every h_k doubled before the carries).
 * 55 products: each unordered pair once; multiplier (2 if i!=j)(2 if i,j odd)(19 if i+j>=10) split into
 * the reference's 32-bit precomputations f0_2.., f5_38, f6_19, f7_38, f8_19, f9_38 (fe.rs:556-571). */
static inline __attribute__((always_inline)) void fe_sq_impl(fe h, const fe f, int dbl) {
  int32_t f2[10], f19[10], f38[10];
  int64_t acc[10];
  for (int i = 0; i < 10; i++) { f2[i] = 2 * f[i]; f19[i] = 19 * f[i]; f38[i] = 38 * f[i]; }
#pragma GCC unroll 10
  for (int k = 0; k < 10; k++) {
    int64_t s = 0;
#pragma GCC unroll 10
    for (int i = 0; i < 10; i++) {
      int j = k - i, wrap = 0;
      if (j < 0) { j += 10; wrap = 1; }
      if (i > j) continue;
      const int oo = (i & 1) && (j & 1);
      const int32_t a = (i != j) ? f2[i] : f[i];
      const int32_t b = wrap ? (oo ? f38[j] : f19[j]) : (oo ? f2[j] : f[j]);
      s += (int64_t)a * (int64_t)b;
    }
    acc[k] = dbl ? s + s : s;
  }
  carry_mul_schedule(h, acc);
}
static void fe_sq(fe h, const fe f) { fe_sq_impl(h, f, 0); }
static void fe_sq2(fe h, const fe f) { fe_sq_impl(h, f, 1); }
static void fe_sqn(fe h, const fe f, int n) { fe_sq(h, f); for (int i = 1; i < n; i++) fe_sq(h, h); }
/* fe.rs:857-944 */
static void fe_invert(fe out, const fe z) {
  fe t0, t1, t2, t3;
  fe_sq(t0, z);
  fe_sqn(t1, t0, 2);
  fe_mul(t1, z, t1);
  fe_mul(t0, t0, t1);
  fe_sq(t2, t0);
  fe_mul(t1, t1, t2);
  fe_sqn(t2, t1, 5);
  fe_mul(t1, t2, t1);
  fe_sqn(t2, t1, 10);
  fe_mul(t2, t2, t1);
  fe_sqn(t3, t2, 20);
  fe_mul(t2, t3, t2);
  fe_sqn(t2, t2, 10);
  fe_mul(t1, t2, t1);
  fe_sqn(t2, t1, 50);
  fe_mul(t2, t2, t1);
  fe_sqn(t3, t2, 100);
  fe_mul(t2, t3, t2);
  fe_sqn(t2, t2, 50);
  fe_mul(t1, t2, t1);
  fe_sqn(t1, t1, 5);
  fe_mul(out, t1, t0);
}
/* fe.rs:946-1035 */
static void fe_pow22523(fe out, const fe z) {
  fe t0, t1, t2;
  fe_sq(t0, z);
  fe_sqn(t1, t0, 2);
  fe_mul(t1, z, t1);
  fe_mul(t0, t0, t1);
  fe_sq(t0, t0);
  fe_mul(t0, t1, t0);
  fe_sqn(t1, t0, 5);
  fe_mul(t0, t1, t0);
  fe_sqn(t1, t0, 10);
  fe_mul(t1, t1, t0);
  fe_sqn(t2, t1, 20);
  fe_mul(t1, t2, t1);
  fe_sqn(t1, t1, 10);
  fe_mul(t0, t1, t0);
  fe_sqn(t1, t0, 50);
  fe_mul(t1, t1, t0);
  fe_sqn(t2, t1, 100);
  fe_mul(t1, t2, t1);
  fe_sqn(t1, t1, 50);
  fe_mul(t0, t1, t0);
  fe_sqn(t0, t0, 2);
  fe_mul(out, t0, z);
}

/* ------------------------------------------------------------------ constants ------------- */
/* constants.rs:56-87 give D, D2, SQRT_M1, BASEEXT as limb literals and constants.rs:89-3738 the
 * BASE table.  They are not copied: they are recomputed here from the curve definition
 * (d = -121665/121666, sqrt(-1) = 2^((p-1)/4), B = (x, 4/5) with x "positive") and compared with the
 * reference's literals by value in tests/test_constants_vs_reference.py. */
static fe D, D2, SQRTM1;
static ge_p3 BASEPT;
static ge_precomp BASE[32][8];       /* BASE[i][j] = (j+1) * 256^i * B, constants.rs:89 */
static int oracle_ready = 0;

static void fe_from_small(fe h, int32_t v) { fe_0(h); h[0] = v; }

/* ------------------------------------------------------------------ group: ge.rs ---------- */
static void p3_0(ge_p3* h) { fe_0(h->X); fe_1(h->Y); fe_1(h->Z); fe_0(h->T); }          /* ge.rs:194-199 */
static void precomp_0(ge_precomp* h) { fe_1(h->ypx); fe_1(h->ymx); fe_0(h->xy2d); }     /* ge.rs:308-312 */
static void cached_0(ge_cached* h) { fe_1(h->YpX); fe_1(h->YmX); fe_1(h->Z); fe_0(h->T2d); } /* ge.rs:338-343 */
/* ge.rs:35-49 */
static void p2_dbl(ge_p1p1* r, const ge_p2* p) {
  fe t0;
  fe_sq(r->X, p->X);
  fe_sq(r->Z, p->Y);
  fe_sq2(r->T, p->Z);
  fe_add(r->Y, p->X, p->Y);
  fe_sq(t0, r->Y);
  fe_add(r->Y, r->Z, r->X);
  fe_sub(r->Z, r->Z, r->X);
  fe_sub(r->X, t0, r->Y);
  fe_sub(r->T, r->T, r->Z);
}
static void p3_dbl(ge_p1p1* r, const ge_p3* p) {                                         /* ge.rs:86-91 */
  ge_p2 q; fe_copy(q.X, p->X); fe_copy(q.Y, p->Y); fe_copy(q.Z, p->Z);
  p2_dbl(r, &q);
}
static void p3_to_cached(ge_cached* r, const ge_p3* p) {                                 /* ge.rs:93-98 */
  fe_add(r->YpX, p->Y, p->X);
  fe_sub(r->YmX, p->Y, p->X);
  fe_copy(r->Z, p->Z);
  fe_mul(r->T2d, p->T, D2);
}
static void p1p1_to_p2(ge_p2* r, const ge_p1p1* p) {                                     /* ge.rs:211-215 */
  fe_mul(r->X, p->X, p->T);
  fe_mul(r->Y, p->Y, p->Z);
  fe_mul(r->Z, p->Z, p->T);
}
static void p1p1_to_p3(ge_p3* r, const ge_p1p1* p) {                                     /* ge.rs:292-297 */
  fe_mul(r->X, p->X, p->T);
  fe_mul(r->Y, p->Y, p->Z);
  fe_mul(r->Z, p->Z, p->T);
  fe_mul(r->T, p->X, p->Y);
}
/* ge.rs:217-233 (sub: ge.rs:235-251 — YpX/YmX swapped, last two lines swapped) */
static void ge_addsub(ge_p1p1* r, const ge_p3* p, const ge_cached* q, int sub) {
  fe t0;
  fe_add(r->X, p->Y, p->X);
  fe_sub(r->Y, p->Y, p->X);
  fe_mul(r->Z, r->X, sub ? q->YmX : q->YpX);
  fe_mul(r->Y, r->Y, sub ? q->YpX : q->YmX);
  fe_mul(r->T, q->T2d, p->T);
  fe_mul(r->X, p->Z, q->Z);
  fe_add(t0, r->X, r->X);
  fe_sub(r->X, r->Z, r->Y);
  fe_add(r->Y, r->Z, r->Y);
  if (!sub) { fe_add(r->Z, t0, r->T); fe_sub(r->T, t0, r->T); }
  else      { fe_sub(r->Z, t0, r->T); fe_add(r->T, t0, r->T); }
}
/* ge.rs:274-290 */
static void ge_madd(ge_p1p1* r, const ge_p3* p, const ge_precomp* q) {
  fe t0;
  fe_add(r->X, p->Y, p->X);
  fe_sub(r->Y, p->Y, p->X);
  fe_mul(r->Z, r->X, q->ypx);
  fe_mul(r->Y, r->Y, q->ymx);
  fe_mul(r->T, q->xy2d, p->T);
  fe_add(t0, p->Z, p->Z);
  fe_sub(r->X, r->Z, r->Y);
  fe_add(r->Y, r->Z, r->Y);
  fe_add(r->Z, t0, r->T);
  fe_sub(r->T, t0, r->T);
}
/* ge.rs:112-122 */
static void p3_tobytes(uint8_t* s, const ge_p3* h) {
  fe recip, x, y;
  fe_invert(recip, h->Z);
  fe_mul(x, h->X, recip);
  fe_mul(y, h->Y, recip);
  fe_tobytes(s, y);
  s[31] ^= (uint8_t)(fe_isnegative(x) << 7);
}
/* ge.rs:124-179; returns 1 on success */
static int p3_frombytes(ge_p3* h, const uint8_t* s) {
  fe u, v, v3, vxx, check;
  fe_frombytes(h->Y, s);
  fe_1(h->Z);
  fe_sq(u, h->Y);
  fe_mul(v, u, D);
  fe_sub(u, u, h->Z);
  fe_add(v, v, h->Z);
  fe_sq(v3, v);
  fe_mul(v3, v3, v);
  fe_sq(h->X, v3);
  fe_mul(h->X, h->X, v);
  fe_mul(h->X, h->X, u);
  fe_pow22523(h->X, h->X);
  fe_mul(h->X, h->X, v3);
  fe_mul(h->X, h->X, u);
  fe_sq(vxx, h->X);
  fe_mul(vxx, vxx, v);
  fe_sub(check, vxx, u);
  if (fe_isnonzero(check)) {
    fe_add(check, vxx, u);
    if (fe_isnonzero(check)) return 0;
    fe_mul(h->X, h->X, SQRTM1);
  }
  if (fe_isnegative(h->X) != (s[31] >> 7)) fe_neg(h->X, h->X);
  fe_mul(h->T, h->X, h->Y);
  return 1;
}
/* ge.rs:411-421 */
static int32_t ct_equal(int32_t b, int32_t c) { uint32_t x = (uint32_t)(b ^ c); x -= 1; return (int32_t)(x >> 31); }
static int32_t ct_negative(int32_t b) { return (b >> 31) & 1; }
/* ge.rs:307-322 */
static void precomp_cmov(ge_precomp* t, const ge_precomp* u, int32_t b) {
  fe_cmov(t->ypx, u->ypx, b); fe_cmov(t->ymx, u->ymx, b); fe_cmov(t->xy2d, u->xy2d, b);
}
static void cached_cmov(ge_cached* t, const ge_cached* u, int32_t b) {
  fe_cmov(t->YpX, u->YpX, b); fe_cmov(t->YmX, u->YmX, b); fe_cmov(t->Z, u->Z, b); fe_cmov(t->T2d, u->T2d, b);
}
/* ge.rs:423-434 */
static void select_precomp(ge_precomp* t, int pos, int32_t b) {
  ge_precomp minus;
  int32_t bneg = ct_negative(b);
  int32_t babs = b - (((-bneg) & b) << 1);
  precomp_0(t);
  for (int i = 0; i < 8; i++) precomp_cmov(t, &BASE[pos][i], ct_equal(babs, i + 1));
  fe_copy(minus.ypx, t->ymx); fe_copy(minus.ymx, t->ypx); fe_neg(minus.xy2d, t->xy2d);
  precomp_cmov(t, &minus, bneg);
}
/* ge.rs:488-500 */
static void select_cached(ge_cached* c, const ge_cached ai[8], int32_t b) {
  ge_cached minus;
  int32_t bneg = ct_negative(b);
  int32_t babs = b - (((-bneg) & b) << 1);
  cached_0(c);
  for (int i = 0; i < 8; i++) cached_cmov(c, &ai[i], ct_equal(babs, i + 1));
  fe_copy(minus.YpX, c->YmX); fe_copy(minus.YmX, c->YpX); fe_copy(minus.Z, c->Z); fe_neg(minus.T2d, c->T2d);
  cached_cmov(c, &minus, bneg);
}
/* ge.rs:443-459 / 519-534: signed radix-16 recoding in i8 arithmetic; e[63] is NOT recentred */
static void recode(int8_t e[64], const uint8_t a[32]) {
  for (int i = 0; i < 32; i++) { e[2 * i] = (int8_t)(a[i] & 15); e[2 * i + 1] = (int8_t)((a[i] >> 4) & 15); }
  int8_t carry = 0;
  for (int i = 0; i < 63; i++) {
    e[i] = (int8_t)(e[i] + carry);
    carry = (int8_t)((e[i] + 8) >> 4);
    e[i] = (int8_t)(e[i] - (carry << 4));
  }
  e[63] = (int8_t)(e[63] + carry);
}
/* ge.rs:442-486 */
static void ge_scalarmult_base(ge_p3* h, const uint8_t a[32]) {
  int8_t e[64];
  ge_precomp t; ge_p1p1 r; ge_p2 s;
  recode(e, a);
  p3_0(h);
  for (int i = 1; i < 64; i += 2) { select_precomp(&t, i / 2, e[i]); ge_madd(&r, h, &t); p1p1_to_p3(h, &r); }
  p3_dbl(&r, h); p1p1_to_p2(&s, &r);
  p2_dbl(&r, &s); p1p1_to_p2(&s, &r);
  p2_dbl(&r, &s); p1p1_to_p2(&s, &r);
  p2_dbl(&r, &s); p1p1_to_p3(h, &r);
  for (int i = 0; i < 64; i += 2) { select_precomp(&t, i / 2, e[i]); ge_madd(&r, h, &t); p1p1_to_p3(h, &r); }
}
/* ge.rs:508-568 */
static void ge_scalarmult(ge_p3* h, const uint8_t a[32], const ge_p3* A) {
  int8_t e[64];
  ge_p1p1 t; ge_p3 u; ge_p2 r; ge_cached c, ai[8];
  recode(e, a);
  p3_to_cached(&ai[0], A);
  for (int i = 0; i < 7; i++) { ge_addsub(&t, A, &ai[i], 0); p1p1_to_p3(&u, &t); p3_to_cached(&ai[i + 1], &u); }
  p3_0(&u);
  select_cached(&c, ai, e[63]);
  ge_addsub(&t, &u, &c, 0);
  for (int i = 62; i >= 0; i--) {
    p1p1_to_p2(&r, &t); p2_dbl(&t, &r);
    p1p1_to_p2(&r, &t); p2_dbl(&t, &r);
    p1p1_to_p2(&r, &t); p2_dbl(&t, &r);
    p1p1_to_p2(&r, &t); p2_dbl(&t, &r);
    p1p1_to_p3(&u, &t);
    select_cached(&c, ai, e[i]);
    ge_addsub(&t, &u, &c, 0);
  }
  p1p1_to_p3(h, &t);
}

/* ------------------------------------------------------------------ SHA-512 (FIPS 180-4) -- */
/* the reference takes SHA-512 from the sha2 crate (Cargo.toml: sha2 = "0.10.6") */
static const uint64_t K512[80] = {
  0x428a2f98d728ae22ULL,0x7137449123ef65cdULL,0xb5c0fbcfec4d3b2fULL,0xe9b5dba58189dbbcULL,0x3956c25bf348b538ULL,0x59f111f1b605d019ULL,0x923f82a4af194f9bULL,0xab1c5ed5da6d8118ULL,
  0xd807aa98a3030242ULL,0x12835b0145706fbeULL,0x243185be4ee4b28cULL,0x550c7dc3d5ffb4e2ULL,0x72be5d74f27b896fULL,0x80deb1fe3b1696b1ULL,0x9bdc06a725c71235ULL,0xc19bf174cf692694ULL,
  0xe49b69c19ef14ad2ULL,0xefbe4786384f25e3ULL,0x0fc19dc68b8cd5b5ULL,0x240ca1cc77ac9c65ULL,0x2de92c6f592b0275ULL,0x4a7484aa6ea6e483ULL,0x5cb0a9dcbd41fbd4ULL,0x76f988da831153b5ULL,
  0x983e5152ee66dfabULL,0xa831c66d2db43210ULL,0xb00327c898fb213fULL,0xbf597fc7beef0ee4ULL,0xc6e00bf33da88fc2ULL,0xd5a79147930aa725ULL,0x06ca6351e003826fULL,0x142929670a0e6e70ULL,
  0x27b70a8546d22ffcULL,0x2e1b21385c26c926ULL,0x4d2c6dfc5ac42aedULL,0x53380d139d95b3dfULL,0x650a73548baf63deULL,0x766a0abb3c77b2a8ULL,0x81c2c92e47edaee6ULL,0x92722c851482353bULL,
  0xa2bfe8a14cf10364ULL,0xa81a664bbc423001ULL,0xc24b8b70d0f89791ULL,0xc76c51a30654be30ULL,0xd192e819d6ef5218ULL,0xd69906245565a910ULL,0xf40e35855771202aULL,0x106aa07032bbd1b8ULL,
  0x19a4c116b8d2d0c8ULL,0x1e376c085141ab53ULL,0x2748774cdf8eeb99ULL,0x34b0bcb5e19b48a8ULL,0x391c0cb3c5c95a63ULL,0x4ed8aa4ae3418acbULL,0x5b9cca4f7763e373ULL,0x682e6ff3d6b2b8a3ULL,
  0x748f82ee5defb2fcULL,0x78a5636f43172f60ULL,0x84c87814a1f0ab72ULL,0x8cc702081a6439ecULL,0x90befffa23631e28ULL,0xa4506cebde82bde9ULL,0xbef9a3f7b2c67915ULL,0xc67178f2e372532bULL,
  0xca273eceea26619cULL,0xd186b8c721c0c207ULL,0xeada7dd6cde0eb1eULL,0xf57d4f7fee6ed178ULL,0x06f067aa72176fbaULL,0x0a637dc5a2c898a6ULL,0x113f9804bef90daeULL,0x1b710b35131c471bULL,
  0x28db77f523047d84ULL,0x32caab7b40c72493ULL,0x3c9ebe0a15c9bebcULL,0x431d67c49c100d4cULL,0x4cc5d4becb3e42b6ULL,0x597f299cfc657e2aULL,0x5fcb6fab3ad6faecULL,0x6c44198c4a475817ULL};
#define ROR(x, n) (((x) >> (n)) | ((x) << (64 - (n))))
typedef struct { uint64_t h[8]; uint8_t buf[128]; size_t fill; uint64_t total; } sha512_t;
static void sha_block(sha512_t* c, const uint8_t* p) {
  uint64_t w[80], a, b, cc, d, e, f, g, h;
  for (int i = 0; i < 16; i++) { w[i] = 0; for (int k = 0; k < 8; k++) w[i] = (w[i] << 8) | p[8 * i + k]; }
  for (int i = 16; i < 80; i++) {
    uint64_t s0 = ROR(w[i - 15], 1) ^ ROR(w[i - 15], 8) ^ (w[i - 15] >> 7);
    uint64_t s1 = ROR(w[i - 2], 19) ^ ROR(w[i - 2], 61) ^ (w[i - 2] >> 6);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  a = c->h[0]; b = c->h[1]; cc = c->h[2]; d = c->h[3]; e = c->h[4]; f = c->h[5]; g = c->h[6]; h = c->h[7];
  for (int i = 0; i < 80; i++) {
    uint64_t t1 = h + (ROR(e, 14) ^ ROR(e, 18) ^ ROR(e, 41)) + ((e & f) ^ (~e & g)) + K512[i] + w[i];
    uint64_t t2 = (ROR(a, 28) ^ ROR(a, 34) ^ ROR(a, 39)) + ((a & b) ^ (a & cc) ^ (b & cc));
    h = g; g = f; f = e; e = d + t1; d = cc; cc = b; b = a; a = t1 + t2;
  }
  c->h[0] += a; c->h[1] += b; c->h[2] += cc; c->h[3] += d; c->h[4] += e; c->h[5] += f; c->h[6] += g; c->h[7] += h;
}
static void sha_init(sha512_t* c) {
  static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL,0xbb67ae8584caa73bULL,0x3c6ef372fe94f82bULL,0xa54ff53a5f1d36f1ULL,
                                 0x510e527fade682d1ULL,0x9b05688c2b3e6c1fULL,0x1f83d9abfb41bd6bULL,0x5be0cd19137e2179ULL};
  memcpy(c->h, iv, sizeof(iv)); c->fill = 0; c->total = 0;
}
static void sha_update(sha512_t* c, const uint8_t* p, size_t n) {
  c->total += n;
  while (n) {
    size_t k = 128 - c->fill; if (k > n) k = n;
    memcpy(c->buf + c->fill, p, k); c->fill += k; p += k; n -= k;
    if (c->fill == 128) { sha_block(c, c->buf); c->fill = 0; }
  }
}
static void sha_final(sha512_t* c, uint8_t out[64]) {
  uint64_t bits = c->total * 8;
  uint8_t pad[256]; size_t n = (c->fill < 112) ? (112 - c->fill) : (240 - c->fill);
  memset(pad, 0, sizeof(pad)); pad[0] = 0x80;
  for (int i = 0; i < 8; i++) pad[n + 8 + i] = (uint8_t)(bits >> (56 - 8 * i));
  uint64_t keep = c->total; sha_update(c, pad, n + 16); c->total = keep;
  for (int i = 0; i < 8; i++) for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(c->h[i] >> (56 - 8 * k));
}

/* ------------------------------------------------------------------ scalars mod L ---------- */
/* scalar.rs:279-744 (sc_mul_add), :759 (sc_add), :1187 (sc_sub), :1596 (sc_mul): ref10's 12 x 21-bit
 * limbs; here restated as: product limbs by loops, then the reference's reduction steps
 * (fold limb i>=12 down with the 21-bit limbs of -L's low part, carry, fold again). */
static const int64_t LM[6] = {666643, 470296, 654183, -997805, 136657, -683901}; /* scalar.rs:~470: s_{i-12..i-7} += s_i * LM */
static void sc_load(int64_t a[12], const uint8_t s[32]) {
  /* 21-bit limbs: limb i = bits [21 i, 21 i + 21); limb 11 keeps everything above bit 231 (scalar.rs:280-291) */
  for (int i = 0; i < 12; i++) {
    int bit = 21 * i; uint64_t v = 0;
    for (int b = 0; b < 5 && (bit >> 3) + b < 32; b++) v |= (uint64_t)s[(bit >> 3) + b] << (8 * b);
    v >>= (bit & 7);
    a[i] = (i < 11) ? (int64_t)(v & 2097151) : (int64_t)v;
  }
}
static void sc_carry_round(int64_t* s, int i) { int64_t c = (s[i] + (1 << 20)) >> 21; s[i + 1] += c; s[i] -= c * (1 << 21); }
static void sc_carry_floor(int64_t* s, int i) { int64_t c = s[i] >> 21; s[i + 1] += c; s[i] -= c * (1 << 21); }
static void sc_fold(int64_t* s, int i) { for (int k = 0; k < 6; k++) s[i - 12 + k] += s[i] * LM[k]; s[i] = 0; }
/* s[0..23] holds the 24-limb value; reduce to 12 limbs and store (scalar.rs:~420-744) */
static void sc_reduce_limbs(uint8_t out[32], int64_t s[25]) {
  for (int i = 0; i <= 22; i += 2) sc_carry_round(s, i);
  for (int i = 1; i <= 21; i += 2) sc_carry_round(s, i);
  for (int i = 23; i >= 18; i--) sc_fold(s, i);
  for (int i = 6; i <= 16; i += 2) sc_carry_round(s, i);
  for (int i = 7; i <= 15; i += 2) sc_carry_round(s, i);
  for (int i = 17; i >= 12; i--) sc_fold(s, i);
  for (int i = 0; i <= 10; i += 2) sc_carry_round(s, i);
  for (int i = 1; i <= 11; i += 2) sc_carry_round(s, i);
  sc_fold(s, 12);
  for (int i = 0; i <= 11; i++) sc_carry_floor(s, i);
  sc_fold(s, 12);
  for (int i = 0; i <= 10; i++) sc_carry_floor(s, i);
  memset(out, 0, 32);
  for (int i = 0; i < 12; i++) {
    int bit = 21 * i; uint64_t v = (uint64_t)s[i] << (bit & 7);
    for (int b = 0; b < 5 && (bit >> 3) + b < 32; b++) out[(bit >> 3) + b] |= (uint8_t)(v >> (8 * b));
  }
}
void orc_sc_muladd(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32]) {
  int64_t A[12], B[12], C[12], s[25];
  sc_load(A, a); sc_load(B, b); sc_load(C, c);
  memset(s, 0, sizeof(s));
  for (int i = 0; i < 12; i++) s[i] = C[i];
  for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) s[i + j] += A[i] * B[j];
  sc_reduce_limbs(out, s);
}
/* Scalar::set_bytes on a 64-byte digest (scalar.rs:175-177 -> integer.rs:386-397): LE integer mod L.
 * The reference uses a big-integer remainder; restated with the same 21-bit machinery (ref10 sc_reduce). */
void orc_sc_reduce64(uint8_t out[32], const uint8_t in[64]) {
  int64_t s[25]; memset(s, 0, sizeof(s));
  for (int i = 0; i < 24; i++) {
    int bit = 21 * i; uint64_t v = 0;
    for (int b = 0; b < 5 && (bit >> 3) + b < 64; b++) v |= (uint64_t)in[(bit >> 3) + b] << (8 * b);
    v >>= (bit & 7);
    s[i] = (i < 23) ? (int64_t)(v & 2097151) : (int64_t)v;
  }
  sc_reduce_limbs(out, s);
}

/* ------------------------------------------------------------------ init ------------------ */
static void fe_from_int_bytes(fe h, const uint8_t s[32]) { fe_frombytes(h, s); }
static void oracle_init(void) {
  if (oracle_ready) return;
  /* d = -121665/121666 */
  fe num, den, inv;
  fe_from_small(num, -121665); fe_from_small(den, 121666);
  fe_invert(inv, den); fe_mul(D, num, inv);
  fe_add(D2, D, D); { fe one; fe_1(one); fe_mul(D2, D2, one); }
  /* sqrt(-1) = 2^((p-1)/4):  (p-1)/4 = 2^253 - 5 = 2*(2^252-3) + 1  =>  2^((p-1)/4) = (2^(2^252-3))^2 * 2 */
  fe two, t; fe_from_small(two, 2);
  fe_pow22523(t, two); fe_sq(t, t); fe_mul(SQRTM1, t, two);
  /* B: y = 4/5, x even */
  uint8_t benc[32]; fe four, five, y;
  fe_from_small(four, 4); fe_from_small(five, 5); fe_invert(inv, five); fe_mul(y, four, inv);
  fe_tobytes(benc, y);
  p3_frombytes(&BASEPT, benc);
  /* BASE[i][j] = (j+1) * 256^i * B as (y+x, y-x, 2dxy)   (constants.rs:89-3738) */
  ge_p3 Pi = BASEPT;                     /* 256^i * B */
  for (int i = 0; i < 32; i++) {
    ge_p3 acc = Pi; ge_cached pc; p3_to_cached(&pc, &Pi);
    for (int j = 0; j < 8; j++) {
      fe recip, x, yy, xy;
      fe_invert(recip, acc.Z); fe_mul(x, acc.X, recip); fe_mul(yy, acc.Y, recip);
      fe_add(BASE[i][j].ypx, yy, x); fe_sub(BASE[i][j].ymx, yy, x);
      fe_mul(xy, x, yy); fe_mul(BASE[i][j].xy2d, xy, D2);
      { fe one; fe_1(one); fe_mul(BASE[i][j].ypx, BASE[i][j].ypx, one); fe_mul(BASE[i][j].ymx, BASE[i][j].ymx, one); }
      ge_p1p1 r; ge_addsub(&r, &acc, &pc, 0); p1p1_to_p3(&acc, &r);
    }
    for (int k = 0; k < 8; k++) { ge_p1p1 r; p3_dbl(&r, &Pi); p1p1_to_p3(&Pi, &r); }
  }
  (void)fe_from_int_bytes;
  oracle_ready = 1;
}
static pthread_once_t once = PTHREAD_ONCE_INIT;
static void ensure(void) { pthread_once(&once, oracle_init); }

/* ------------------------------------------------------------------ exported API ---------- */
static void ext_in(ge_p3* p, const int32_t e[40]) { memcpy(p->X, e, 40); memcpy(p->Y, e + 10, 40); memcpy(p->Z, e + 20, 40); memcpy(p->T, e + 30, 40); }
static void ext_out(int32_t e[40], const ge_p3* p) { memcpy(e, p->X, 40); memcpy(e + 10, p->Y, 40); memcpy(e + 20, p->Z, 40); memcpy(e + 30, p->T, 40); }

void orc_init(void) { ensure(); }
/* Point::mul(s, None) + marshal_binary  (point.rs:207-213, 35-41) */
void orc_mul_base(uint8_t out_enc[32], int32_t out_ext[40], const uint8_t scalar[32]) {
  ensure(); ge_p3 h; ge_scalarmult_base(&h, scalar);
  if (out_enc) p3_tobytes(out_enc, &h);
  if (out_ext) ext_out(out_ext, &h);
}
/* Point::mul(s, Some(P)) + marshal_binary  (point.rs:214-220) */
void orc_mul(uint8_t out_enc[32], int32_t out_ext[40], const uint8_t scalar[32], const int32_t pt_ext[40]) {
  ensure(); ge_p3 A, h; ext_in(&A, pt_ext); ge_scalarmult(&h, scalar, &A);
  if (out_enc) p3_tobytes(out_enc, &h);
  if (out_ext) ext_out(out_ext, &h);
}
/* unmarshal_binary (point.rs:43-50) */
int orc_decode(int32_t out_ext[40], const uint8_t enc[32]) { ensure(); ge_p3 h; memset(&h, 0, sizeof(h)); int ok = p3_frombytes(&h, enc); ext_out(out_ext, &h); return ok; }
void orc_encode(uint8_t enc[32], const int32_t ext[40]) { ensure(); ge_p3 h; ext_in(&h, ext); p3_tobytes(enc, &h); }
/* Point::add / sub / neg  (point.rs:179-204) */
void orc_add(int32_t out[40], const int32_t a[40], const int32_t b[40], int sub) {
  ensure(); ge_p3 A, B, R; ge_cached c; ge_p1p1 r; ext_in(&A, a); ext_in(&B, b);
  p3_to_cached(&c, &B); ge_addsub(&r, &A, &c, sub); p1p1_to_p3(&R, &r); ext_out(out, &R);
}
void orc_neg(int32_t out[40], const int32_t a[40]) { ge_p3 A; ext_in(&A, a); fe_neg(A.X, A.X); fe_neg(A.T, A.T); ext_out(out, &A); }
void orc_base(int32_t out[40]) { ensure(); ext_out(out, &BASEPT); }
void orc_null(int32_t out[40]) { ge_p3 z; p3_0(&z); ext_out(out, &z); }
/* Point::embed(data, rand) / Point::pick(rand) = embed(None, rand)  (point.rs:90-92, 106-167) over a REPLAYED key stream: block i of
 * `stream` is what the i-th `rand.xor_key_stream(&mut b, &[0; 32])` of the loop yields.  data_len < 0 means `None`.
 *   dl = min(embed_len() = 29, data.len())            (longer data is silently truncated, :108-116)
 *   candidate b = block; with data: b[0] = dl, b[1..1+dl] = data[..dl]   (:121-127)
 *   !set_bytes(b)                         -> next block                  (:129-132)
 *   None:  Q = mul(COFACTOR_SCALAR = 8, P); Q == NULL_POINT (by encoding, point.rs:227) -> next block, else return Q   (:144-153)
 *   Some:  mul(PRIME_ORDER_SCALAR = L, P) == NULL_POINT -> return P (the decoded candidate itself), else next block   (:158-164)
 * Both constants are the unreduced 32-byte integers (constants.rs:45-49).  Returns the number of blocks consumed, or -1 when the
 * stream ran out before a candidate was accepted (the reference would keep drawing). */
long orc_embed(int32_t out_ext[40], uint8_t out_enc[32], const uint8_t* data, long data_len, const uint8_t* stream, size_t blocks) {
  ensure();
  static const uint8_t Lb[32] = {0xed,0xd3,0xf5,0x5c,0x1a,0x63,0x12,0x58,0xd6,0x9c,0xf7,0xa2,0xde,0xf9,0xde,0x14,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0x10};
  static const uint8_t eight[32] = {8}, null_enc[32] = {1};
  size_t dl = (255 - 8 - 8) / 8;
  const size_t have = data_len < 0 ? 0 : (size_t)data_len;
  if (dl > have) dl = have;
  for (size_t i = 0; i < blocks; i++) {
    uint8_t b[32], enc[32];
    memcpy(b, stream + 32 * i, 32);
    if (data_len >= 0) { b[0] = (uint8_t)dl; memcpy(b + 1, data, dl); }
    ge_p3 P, Q;
    if (!p3_frombytes(&P, b)) continue;
    ge_scalarmult(&Q, data_len < 0 ? eight : Lb, &P);
    p3_tobytes(enc, &Q);
    const int is_null = memcmp(enc, null_enc, 32) == 0;
    if (data_len < 0 ? is_null : !is_null) continue;
    const ge_p3* R = data_len < 0 ? &Q : &P;
    if (out_ext) ext_out(out_ext, R);
    if (out_enc) p3_tobytes(out_enc, R);
    return (long)i + 1;
  }
  return -1;
}
/* Point::data (point.rs:169-177): the bytes embed() placed; -1 = PointError::EmbedDataLength */
long orc_point_data(uint8_t out[29], const int32_t ext[40]) {
  ensure(); ge_p3 h; ext_in(&h, ext); uint8_t b[32]; p3_tobytes(b, &h);
  const size_t dl = b[0];
  if (dl > (255 - 8 - 8) / 8) return -1;
  memcpy(out, b + 1, dl);
  return (long)dl;
}
/* read-only views for the constants-vs-reference test: D, D2, SQRT_M1, BASE[i][j] as 32-byte values */
void orc_const_bytes(uint8_t out[32], int which) { ensure(); fe_tobytes(out, which == 0 ? D : which == 1 ? D2 : SQRTM1); }
void orc_base_table_bytes(uint8_t out[96], int i, int j) { ensure(); fe_tobytes(out, BASE[i][j].ypx); fe_tobytes(out + 32, BASE[i][j].ymx); fe_tobytes(out + 64, BASE[i][j].xy2d); }
void orc_sha512(uint8_t out[64], const uint8_t* msg, size_t n) { sha512_t c; sha_init(&c); sha_update(&c, msg, n); sha_final(&c, out); }
/* Scalar::marshal_binary: value mod L (scalar.rs:91-100) */
void orc_sc_reduce32(uint8_t out[32], const uint8_t in[32]) { uint8_t w[64]; memset(w, 0, 64); memcpy(w, in, 32); orc_sc_reduce64(out, w); }

/* schnorr::sign with the nonce k supplied (schnorr_sig.rs:25-47, hash :128-141):
 *   R = k*B, A = x*B, h = SHA-512(enc R || enc A || msg) mod L, s = k + x*h, out = enc R || (s mod L) */
void orc_schnorr_sign(uint8_t sig[64], const uint8_t x[32], const uint8_t k[32], const uint8_t* msg, size_t n) {
  ensure();
  uint8_t Renc[32], Aenc[32], dig[64], h[32], s[32];
  orc_mul_base(Renc, NULL, k);
  orc_mul_base(Aenc, NULL, x);
  sha512_t c; sha_init(&c); sha_update(&c, Renc, 32); sha_update(&c, Aenc, 32); sha_update(&c, msg, n); sha_final(&c, dig);
  orc_sc_reduce64(h, dig);
  orc_sc_muladd(s, x, h, k);              /* xh = x*h (sc_mul), s = k + xh (sc_add); both canonical */
  memcpy(sig, Renc, 32);
  orc_sc_reduce32(sig + 32, s);           /* marshal_to -> marshal_binary reduces again (scalar.rs:91-100) */
}
/* Curve::new_key_and_seed_with_input (curve.rs:74-87): secret = clamp(SHA-512(seed)[0..32]) unreduced, prefix = [32..64] */
void orc_eddsa_expand(uint8_t secret[32], uint8_t prefix[32], uint8_t pub[32], const uint8_t seed[32]) {
  uint8_t d[64]; orc_sha512(d, seed, 32);
  d[0] &= 0xf8; d[31] &= 0x7f; d[31] |= 0x40;
  memcpy(secret, d, 32); memcpy(prefix, d + 32, 32);
  if (pub) orc_mul_base(pub, NULL, secret);
}
/* EdDSA::sign (eddsa_sig.rs:120-152): r = SHA-512(prefix || msg) mod L, then the Schnorr equations */
void orc_eddsa_sign(uint8_t sig[64], const uint8_t seed[32], const uint8_t* msg, size_t n) {
  uint8_t secret[32], prefix[32], dig[64], r[32];
  orc_eddsa_expand(secret, prefix, NULL, seed);
  sha512_t c; sha_init(&c); sha_update(&c, prefix, 32); sha_update(&c, msg, n); sha_final(&c, dig);
  orc_sc_reduce64(r, dig);
  orc_schnorr_sign(sig, secret, r, msg, n);
}

/* PubPoly::eval (share/poly.rs:457-469): xi = set_int64(1 + i); v = null; for j = t-1..0: v = v.mul(xi, Some(v)); v = v.add(v, commits[j]) */
void orc_pubpoly_eval(uint8_t out_enc[32], const int32_t* commits_ext, size_t t, uint32_t index) {
  ensure();
  uint8_t xi[32]; memset(xi, 0, 32);
  uint64_t x = (uint64_t)index + 1;
  for (int b = 0; b < 8; b++) xi[b] = (uint8_t)(x >> (8 * b));
  ge_p3 v, m, c; ge_cached cc; ge_p1p1 r;
  p3_0(&v);
  for (size_t j = t; j-- > 0;) {
    ge_scalarmult(&m, xi, &v);
    ext_in(&c, commits_ext + 40 * j);
    p3_to_cached(&cc, &c); ge_addsub(&r, &m, &cc, 0); p1p1_to_p3(&v, &r);
  }
  p3_tobytes(out_enc, &v);
}

/* PriPoly::eval (share/poly.rs:133-141): xi = set_int64(1 + i); v = zero; for j = t-1..0: v = v * xi (sc_mul); v = v + coeffs[j] (sc_add).
 * Both steps leave the canonical residue, so one sc_mul_add per coefficient gives the same 32 bytes. */
void orc_pripoly_eval(uint8_t out[32], const uint8_t* coeffs, size_t t, uint32_t index) {
  uint8_t xi[32], v[32], w[32];
  memset(xi, 0, 32); memset(v, 0, 32);
  uint64_t x = (uint64_t)index + 1;
  for (int b = 0; b < 8; b++) xi[b] = (uint8_t)(x >> (8 * b));
  for (size_t j = t; j-- > 0;) { orc_sc_muladd(w, v, xi, coeffs + 32 * j); memcpy(v, w, 32); }
  memcpy(out, v, 32);
}

/* recover_commit's accumulation (share/poly.rs:579-600): acc = null; for each i: tmp = mul(c_i, Some(y_i)); acc = add(acc, tmp).
 * (The Lagrange coefficients c_i are scalar arithmetic and are the caller's; PubPoly::add / recover_pub_poly, poly.rs:486-507,
 * 607-634, reduce to the same sum per coefficient.) */
void orc_lincomb(uint8_t out_enc[32], const uint8_t* scalars, const int32_t* pts_ext, size_t t) {
  ensure();
  ge_p3 acc, p, m; ge_cached cc; ge_p1p1 r;
  p3_0(&acc);
  for (size_t i = 0; i < t; i++) {
    ext_in(&p, pts_ext + 40 * i);
    ge_scalarmult(&m, scalars + 32 * i, &p);
    p3_to_cached(&cc, &m); ge_addsub(&r, &acc, &cc, 0); p1p1_to_p3(&acc, &r);
  }
  p3_tobytes(out_enc, &acc);
}

/* ---- verification ---------------------------------------------------------------------------
 * status codes (this repo's numbering of the reference's SignatureError variants):
 *   0 valid, 1 InvalidSignatureLength, 2 SignatureNotCanonical, 3 RNotCanonical, 4 R does not decode
 *   (MarshallingError), 5 RSmallOrder, 6 PublicKeyNotCanonical, 7 public key does not decode,
 *   8 PublicKeySmallOrder, 9 InvalidSignature (equation fails) */
/* scalar.rs:54-75 */
static int sc_is_canonical(const uint8_t sb[32]) {
  static const uint8_t Lb[32] = {0xed,0xd3,0xf5,0x5c,0x1a,0x63,0x12,0x58,0xd6,0x9c,0xf7,0xa2,0xde,0xf9,0xde,0x14,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0x10};
  if ((sb[31] & 0xf0) == 0) return 1;
  uint8_t c = 0, n = 1;
  for (int i = 31; i >= 0; i--) {
    c |= (uint8_t)((((uint16_t)sb[i] - (uint16_t)Lb[i]) >> 8) & n);
    n &= (uint8_t)((((uint16_t)sb[i] ^ (uint16_t)Lb[i]) - 1) >> 8);
  }
  return c != 0;
}
/* point.rs:315-337, expression for expression (u16 wrapping arithmetic: Cargo.toml:9-10 sets overflow-checks = false).
 * NB the reference writes  d = (0xED - (1 - b0)) >> 8  where libsodium, which it cites, has (0xED - 1 - b0) >> 8:
 * with bytes 1..30 = 0xff and b31 & 0x7f = 0x7f it therefore reports "not canonical" for every b0 >= 0x14, i.e. also
 * for the 217 canonical values y = p-217 .. p-1, and that is what a drop-in has to answer too. */
static int pt_is_canonical(const uint8_t b[32]) {
  uint8_t c = (uint8_t)((b[31] & 0x7f) ^ 0x7f);
  for (int i = 30; i >= 1; i--) c |= (uint8_t)(b[i] ^ 0xff);
  c = (uint8_t)(((uint16_t)((uint16_t)c - 1u)) >> 8);
  uint16_t inner = (uint16_t)(1u - (uint16_t)b[0]);
  uint8_t d = (uint8_t)(((uint16_t)(0xEDu - inner)) >> 8);
  return 1 - (c & d & 1) == 1;
}
/* point.rs:286-313 with WEAK_KEYS (constants.rs:3744-3775) regenerated: the encodings of the points of
 * order 4 (y=0), 1 (y=1), 8 (two values of y) and 2 (y=p-1); the order-8 y are +-sqrt of ... computed
 * at init from the group law (8-torsion = the points killed by 8) */
static uint8_t WEAK[5][32];
static int weak_ready = 0;
static void weak_init(void) {
  if (weak_ready) return;
  memset(WEAK, 0, sizeof(WEAK));
  WEAK[1][0] = 1;                                      /* y = 1 */
  memset(WEAK[4], 0xff, 32); WEAK[4][0] = 0xec; WEAK[4][31] = 0x7f;   /* y = p - 1 */
  /* order-8 points: find T with 4T = (0,-1)... simplest: scan small y for a point P with 8P = O and 4P != O */
  int found = 0;
  for (uint32_t y = 2; found < 1; y++) {
    uint8_t e[32]; memset(e, 0, 32); e[0] = (uint8_t)y; e[1] = (uint8_t)(y >> 8); e[2] = (uint8_t)(y >> 16);
    ge_p3 P; if (!p3_frombytes(&P, e)) continue;
    /* Q = L * P has order dividing 8 */
    uint8_t Lb[32] = {0xed,0xd3,0xf5,0x5c,0x1a,0x63,0x12,0x58,0xd6,0x9c,0xf7,0xa2,0xde,0xf9,0xde,0x14,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0x10};
    ge_p3 Q; ge_scalarmult(&Q, Lb, &P);
    uint8_t four[32]; memset(four, 0, 32); four[0] = 4;
    ge_p3 Q4; ge_scalarmult(&Q4, four, &Q);
    uint8_t enc4[32]; p3_tobytes(enc4, &Q4);
    if (memcmp(enc4, WEAK[4], 32) != 0) continue;      /* 4Q == (0,-1)  <=>  Q has order 8 */
    uint8_t q[32]; p3_tobytes(q, &Q); q[31] &= 0x7f;
    memcpy(WEAK[2], q, 32);
    /* the other order-8 y is p - y */
    fe fy, fny; fe_frombytes(fy, q); fe_neg(fny, fy); fe_tobytes(WEAK[3], fny);
    if (memcmp(WEAK[2], WEAK[3], 32) > 0) { uint8_t t[32]; memcpy(t, WEAK[2], 32); memcpy(WEAK[2], WEAK[3], 32); memcpy(WEAK[3], t, 32); }
    found = 1;
  }
  weak_ready = 1;
}
void orc_weak_keys(uint8_t out[160]) { ensure(); weak_init(); memcpy(out, WEAK, 160); }
static int pt_has_small_order(const ge_p3* P) {
  uint8_t s[32]; p3_tobytes(s, P);
  uint8_t c[5] = {0, 0, 0, 0, 0};
  for (int j = 0; j < 31; j++) for (int i = 0; i < 5; i++) c[i] |= (uint8_t)(s[j] ^ WEAK[i][j]);
  for (int i = 0; i < 5; i++) c[i] |= (uint8_t)((s[31] & 0x7f) ^ WEAK[i][31]);
  uint16_t k = 0;
  for (int i = 0; i < 5; i++) k |= (uint16_t)((uint16_t)c[i] - 1);
  return ((k >> 8) & 1) > 0;
}
/* PointCanCheckCanonicalAndSmallOrder (group.rs:71-78) on received bytes: bit 0 = is_canonical(b) (point.rs:315-337), bit 1 = has_small_order()
 * of the point the bytes unmarshal to (point.rs:286-313, on its marshal_binary), bit 2 = the bytes unmarshal at all (point.rs:43-50) */
int orc_point_checks(const uint8_t enc[32]) {
  ensure(); weak_init();
  ge_p3 P;
  const int dec = p3_frombytes(&P, enc);
  return pt_is_canonical(enc) | ((dec && pt_has_small_order(&P)) ? 2 : 0) | (dec ? 4 : 0);
}
/* the same on a point the caller holds: both checks on its own marshal_binary */
int orc_point_checks_ext(const int32_t ext[40]) {
  ensure(); weak_init();
  ge_p3 P; ext_in(&P, ext);
  uint8_t enc[32]; p3_tobytes(enc, &P);
  return pt_is_canonical(enc) | (pt_has_small_order(&P) ? 2 : 0);
}
/* flavor 0: eddsa::verify_with_checks (eddsa_sig.rs:159-212); 1: schnorr::verify_with_checks (schnorr_sig.rs:53-110) */
int orc_verify(int flavor, const uint8_t pub[32], const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len) {
  ensure(); weak_init();
  if (sig_len != 64) return 1;
  ge_p3 R, A;
  if (flavor == 0) {
    if (!sc_is_canonical(sig + 32)) return 2;
    if (!pt_is_canonical(sig)) return 3;
    if (!p3_frombytes(&R, sig)) return 4;
    if (pt_has_small_order(&R)) return 5;
    if (!pt_is_canonical(pub)) return 6;
    if (!p3_frombytes(&A, pub)) return 7;
    if (pt_has_small_order(&A)) return 8;
  } else {
    if (!p3_frombytes(&R, sig)) return 4;
    if (!pt_is_canonical(sig)) return 3;
    if (pt_has_small_order(&R)) return 5;
    if (!sc_is_canonical(sig + 32)) return 2;
    if (!p3_frombytes(&A, pub)) return 7;
    if (!pt_is_canonical(pub)) return 6;
    if (pt_has_small_order(&A)) return 8;
  }
  uint8_t dig[64], h[32], Renc[32], Aenc[32];
  if (flavor == 0) { memcpy(Renc, sig, 32); memcpy(Aenc, pub, 32); }      /* raw bytes, eddsa_sig.rs:194-197 */
  else { p3_tobytes(Renc, &R); p3_tobytes(Aenc, &A); }                    /* marshal_to of the decoded points, schnorr_sig.rs:128-141 */
  sha512_t c; sha_init(&c); sha_update(&c, Renc, 32); sha_update(&c, Aenc, 32); sha_update(&c, msg, n); sha_final(&c, dig);
  orc_sc_reduce64(h, dig);
  ge_p3 S, hA, RhA; ge_cached cc; ge_p1p1 r;
  ge_scalarmult_base(&S, sig + 32);
  ge_scalarmult(&hA, h, &A);
  p3_to_cached(&cc, &hA); ge_addsub(&r, &R, &cc, 0); p1p1_to_p3(&RhA, &r);
  uint8_t e1[32], e2[32]; p3_tobytes(e1, &RhA); p3_tobytes(e2, &S);
  return memcmp(e1, e2, 32) == 0 ? 0 : 9;
}

/* ---- batches, optionally multi-threaded (CPU baseline: 1 core = the faithful comparison) ---- */
typedef struct { int kind; int flavor; size_t lo, hi; const uint8_t* sc; const int32_t* pts; const uint8_t* k; const uint8_t* msgs; const uint32_t* off; uint8_t* out; } job_t;
static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  for (size_t i = j->lo; i < j->hi; i++) {
    if (j->kind == 0) orc_mul_base(j->out + 32 * i, NULL, j->sc + 32 * i);
    else if (j->kind == 1) orc_mul(j->out + 32 * i, NULL, j->sc + 32 * i, j->pts + 40 * i);
    else if (j->kind == 2) orc_schnorr_sign(j->out + 64 * i, j->sc + 32 * i, j->k + 32 * i, j->msgs + j->off[i], j->off[i + 1] - j->off[i]);
    else if (j->kind == 3) j->out[i] = (uint8_t)orc_verify(j->flavor, j->sc + 32 * i, j->msgs + j->off[i], j->off[i + 1] - j->off[i], j->k + 64 * i, 64);
    else if (j->kind == 4) {      /* a received point: unmarshal_binary (point.rs:43-51), then mul; an encoding that does not decode -> ok = 0, neutral element out */
      int32_t ext[40];
      int ok = orc_decode(ext, j->k + 32 * i);
      if (!ok) orc_null(ext);
      orc_mul(j->out + 32 * i, NULL, j->sc + 32 * i, ext);
      if (j->msgs) ((uint8_t*)j->msgs)[i] = (uint8_t)ok;
    }
    else orc_encode(j->out + 32 * i, j->pts + 40 * i);      /* kind 5: marshal_binary */
  }
  return NULL;
}
static void run_batch(job_t proto, size_t n, int nthreads) {
  ensure();
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
  job_t* jobs = (job_t*)malloc(sizeof(job_t) * nthreads);
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = proto; jobs[t].lo = n * t / nthreads; jobs[t].hi = n * (t + 1) / nthreads;
    if (nthreads == 1) worker(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
}
void orc_mul_base_batch(uint8_t* out_enc, const uint8_t* scalars, size_t n, int nthreads) {
  job_t j; memset(&j, 0, sizeof(j)); j.kind = 0; j.sc = scalars; j.out = out_enc; run_batch(j, n, nthreads);
}
void orc_mul_batch(uint8_t* out_enc, const uint8_t* scalars, const int32_t* pts_ext, size_t n, int nthreads) {
  job_t j; memset(&j, 0, sizeof(j)); j.kind = 1; j.sc = scalars; j.pts = pts_ext; j.out = out_enc; run_batch(j, n, nthreads);
}
void orc_mul_enc_batch(uint8_t* out_enc, uint8_t* ok, const uint8_t* scalars, const uint8_t* pts_enc, size_t n, int nthreads) {
  job_t j; memset(&j, 0, sizeof(j)); j.kind = 4; j.sc = scalars; j.k = pts_enc; j.msgs = ok; j.out = out_enc; run_batch(j, n, nthreads);
}
void orc_encode_batch(uint8_t* out_enc, const int32_t* pts_ext, size_t n, int nthreads) {
  job_t j; memset(&j, 0, sizeof(j)); j.kind = 5; j.pts = pts_ext; j.out = out_enc; run_batch(j, n, nthreads);
}
void orc_schnorr_sign_batch(uint8_t* sigs, const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* off, size_t n, int nthreads) {
  job_t j; memset(&j, 0, sizeof(j)); j.kind = 2; j.sc = x; j.k = k; j.msgs = msgs; j.off = off; j.out = sigs; run_batch(j, n, nthreads);
}
void orc_verify_batch(uint8_t* status, int flavor, const uint8_t* pubs, const uint8_t* msgs, const uint32_t* off, const uint8_t* sigs, size_t n, int nthreads) {
  weak_init();
  job_t j; memset(&j, 0, sizeof(j)); j.kind = 3; j.flavor = flavor; j.sc = pubs; j.k = sigs; j.msgs = msgs; j.off = off; j.out = status; run_batch(j, n, nthreads);
}
