// C++ mirror of the public-polynomial operations of kyber-rs that reach the engine.
//   PriPoly::commit   /root/reference src/share/poly.rs:195-206   t x mul(coeff, b)  -> one batch
//   PubPoly::eval     src/share/poly.rs:457-469                   Horner with x = i + 1
//   PubPoly::shares   src/share/poly.rs:472-478                   eval at 0..n-1 -> one batch
//   PubPoly::check    src/share/poly.rs:526-530                   eval(s.i) == mul(s.v, b)
//   PubPoly::add / equal   src/share/poly.rs:486-523              one add batch / one eq batch
//   recover_commit    src/share/poly.rs:566-603                   Lagrange at 0: one linear combination
//   recover_pub_poly  src/share/poly.rs:607-634                   t linear combinations over shared points
//   PriPoly::mul, minus_const, lagrange_basis   src/share/poly.rs:213-235, 313-319, 640-668   scalar side, host
//   PriPoly::shares, recover_secret             src/share/poly.rs:144-152, 244-280              kyb_pripoly_eval_batch / kyb_lagrange_coeffs_batch + host dot product
#pragma once
#include <algorithm>
#include <optional>
#include <stdexcept>
#include <vector>

#include "edwards25519.hpp"

namespace kyber {
namespace share {

using group::edwards25519::Point;
using group::edwards25519::Scalar;

struct PubShare { size_t i; Point v; };
struct PriShare { size_t i; Scalar v; };

// poly.rs:671-688
struct PolyError : std::runtime_error { using std::runtime_error::runtime_error; };

class PubPoly {
 public:
  std::optional<Point> b;        // base point (None = the standard base)
  std::vector<Point> commits;

  size_t threshold() const { return commits.size(); }
  Point commit() const { return commits[0]; }

  std::vector<PubShare> eval_many(const std::vector<uint32_t>& idx) const {
    std::vector<int32_t> c(40 * commits.size()), out(40 * idx.size());
    for (size_t j = 0; j < commits.size(); ++j) std::memcpy(&c[40 * j], commits[j].limbs(), 160);
    group::edwards25519::detail::engine_must(
        kyb_pubpoly_eval_batch(c.data(), commits.size(), idx.data(), idx.size(), nullptr, out.data()), "PubPoly::eval");
    std::vector<PubShare> r(idx.size());
    for (size_t k = 0; k < idx.size(); ++k) { r[k].i = idx[k]; std::memcpy(r[k].v.ge, &out[40 * k], 160); }
    return r;
  }
  PubShare eval(size_t i) const { return eval_many({(uint32_t)i})[0]; }
  std::vector<PubShare> shares(size_t n) const {
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    return eval_many(idx);
  }
  bool check(const PriShare& s) const {
    PubShare pv = eval(s.i);
    Point ps = Point().mul(s.v, b ? &*b : nullptr);
    return pv.v == ps;
  }
  // poly.rs:486-507
  PubPoly add(const PubPoly& q) const {
    if (threshold() != q.threshold()) throw PolyError("different number of coefficients");
    const size_t t = commits.size();
    std::vector<int32_t> a(40 * t), c(40 * t), out(40 * t);
    for (size_t j = 0; j < t; ++j) { std::memcpy(&a[40 * j], commits[j].limbs(), 160); std::memcpy(&c[40 * j], q.commits[j].limbs(), 160); }
    group::edwards25519::detail::engine_must(kyb_add_batch(a.data(), c.data(), t, out.data(), 0), "PubPoly::add");
    PubPoly r;
    r.b = b;
    r.commits.resize(t);
    for (size_t j = 0; j < t; ++j) std::memcpy(r.commits[j].ge, &out[40 * j], 160);
    return r;
  }
  // poly.rs:511-523 (compares the first threshold() commitments of both sides)
  bool equal(const PubPoly& q) const {
    const size_t t = commits.size();
    if (q.commits.size() < t) throw std::out_of_range("PubPoly::equal: q has fewer commitments");   // the reference indexes q.commits[i] and panics
    std::vector<int32_t> a(40 * t), c(40 * t);
    std::vector<uint8_t> eq(t);
    for (size_t j = 0; j < t; ++j) { std::memcpy(&a[40 * j], commits[j].limbs(), 160); std::memcpy(&c[40 * j], q.commits[j].limbs(), 160); }
    group::edwards25519::detail::engine_must(kyb_equal_batch(a.data(), c.data(), t, eq.data()), "PubPoly::equal");
    bool all = true;
    for (uint8_t e : eq) all &= e != 0;
    return all;
  }
};

class PriPoly {
 public:
  std::vector<Scalar> coeffs;
  size_t threshold() const { return coeffs.size(); }
  // poly.rs:133-141: Horner in the scalar field, x = i + 1
  PriShare eval(size_t i) const {
    Scalar xi = Scalar().set_int64(1 + (int64_t)i), v = Scalar().zero();
    for (size_t j = coeffs.size(); j-- > 0;) v = v * xi + coeffs[j];
    return PriShare{i, v};
  }
  // poly.rs:144-152: the n shares p(1) .. p(n) — one engine call (kyb_pripoly_eval_batch: constant time in the coefficients) instead of
  // n * t Scalar multiply-adds; same canonical residues as eval() above (tests compare the two)
  std::vector<PriShare> shares(size_t n) const {
    std::vector<uint8_t> c(32 * coeffs.size()), out(32 * n);
    for (size_t j = 0; j < coeffs.size(); ++j) std::memcpy(&c[32 * j], coeffs[j].v.data(), 32);
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    group::edwards25519::detail::engine_must(kyb_pripoly_eval_batch(c.data(), 1, coeffs.size(), idx.data(), n, out.data()), "PriPoly::shares");
    std::vector<PriShare> r(n);
    for (size_t i = 0; i < n; ++i) { r[i].i = i; std::memcpy(r[i].v.v.data(), &out[32 * i], 32); }
    for (auto& b : c) { volatile uint8_t* q = &b; *q = 0; }
    for (auto& b : out) { volatile uint8_t* q = &b; *q = 0; }
    return r;
  }
  // poly.rs:213-235
  PriPoly mul(const PriPoly& q) const {
    PriPoly r;
    r.coeffs.assign(coeffs.size() + q.coeffs.size() - 1, Scalar().zero());
    for (size_t i = 0; i < coeffs.size(); ++i)
      for (size_t j = 0; j < q.coeffs.size(); ++j) r.coeffs[i + j] = r.coeffs[i + j] + coeffs[i] * q.coeffs[j];
    return r;
  }
  // poly.rs:195-206
  // Every in-tree caller passes Some(base) (vss.rs:303: the suite's generator), which sends the reference down the
  // variable-base routine with P = B.  When the limbs handed in are the generator's own (as Point::base() returns
  // them) the fixed-base kernel gives the same points (same affine normal form) at a quarter of the call time.
  PubPoly commit(const Point* base) const {
    PubPoly p;
    if (base) {
      p.b = *base;
      static const Point generator = Point().base();
      if (std::memcmp(base->limbs(), generator.limbs(), sizeof(generator.ge)) == 0) {
        p.commits = Point::mul_batch(coeffs, nullptr);
        return p;
      }
      std::vector<Point> bs(coeffs.size(), *base);
      p.commits = Point::mul_batch(coeffs, &bs);
    } else {
      p.commits = Point::mul_batch(coeffs, nullptr);
    }
    return p;
  }
};

// Verifier's side of a DKG round (vss/pedersen/vss.rs:904-909 for every dealer): polynomial g evaluated at idx[g],
// all in one launch.  The polynomials must share one threshold.
inline std::vector<PubShare> eval_each(const std::vector<PubPoly>& polys, const std::vector<uint32_t>& idx) {
  const size_t m = polys.size();
  if (idx.size() != m) throw std::invalid_argument("eval_each: one index per polynomial");
  if (m == 0) return {};
  const size_t t = polys[0].threshold();
  std::vector<int32_t> c(40 * t * m), out(40 * m);
  for (size_t g = 0; g < m; ++g) {
    if (polys[g].threshold() != t) throw PolyError("different number of coefficients");
    for (size_t j = 0; j < t; ++j) std::memcpy(&c[40 * (g * t + j)], polys[g].commits[j].limbs(), 160);
  }
  group::edwards25519::detail::engine_must(kyb_pubpoly_eval_multi_batch(c.data(), t, m, idx.data(), 1, nullptr, out.data()), "eval_each");
  std::vector<PubShare> r(m);
  for (size_t g = 0; g < m; ++g) { r[g].i = idx[g]; std::memcpy(r[g].v.ge, &out[40 * g], 160); }
  return r;
}

// The distributed public polynomial of a DKG round (dkg.rs:905-953 folds PubPoly::add over the dealers): the
// coefficient-wise sum of all polynomials in one launch.
inline PubPoly sum_polys(const std::vector<PubPoly>& polys) {
  if (polys.empty()) throw std::invalid_argument("sum_polys: no polynomial");
  const size_t n = polys.size(), t = polys[0].threshold();
  std::vector<int32_t> c(40 * t * n), out(40 * t);
  for (size_t d = 0; d < n; ++d) {
    if (polys[d].threshold() != t) throw PolyError("different number of coefficients");
    for (size_t j = 0; j < t; ++j) std::memcpy(&c[40 * (j * n + d)], polys[d].commits[j].limbs(), 160);     // coefficient-major
  }
  group::edwards25519::detail::engine_must(kyb_sum_batch(c.data(), t, n, nullptr, out.data()), "sum_polys");
  PubPoly r;
  r.b = polys[0].b;
  r.commits.resize(t);
  for (size_t j = 0; j < t; ++j) std::memcpy(r.commits[j].ge, &out[40 * j], 160);
  return r;
}

// The same two from the wire: the commitments of m deals as they arrive (Deal.commitments, vss/pedersen/vss.rs:113-124: t
// 32-byte encodings per dealer, dealer after dealer).  The reference unmarshals every one on the CPU (point.rs:43-51) before
// it can evaluate or add; here the encodings go to the GPU as they are.  A commitment that does not decode raises the
// reference's unmarshal error for its dealer.
inline void wire_check(const std::vector<uint8_t>& commits_enc, size_t t) {
  if (t == 0 || commits_enc.size() % (32 * t) != 0) throw std::invalid_argument("wire commitments: t encodings of 32 bytes per dealer");
}
inline std::vector<PubShare> eval_each_wire(const std::vector<uint8_t>& commits_enc, size_t t, const std::vector<uint32_t>& idx) {
  wire_check(commits_enc, t);
  const size_t m = commits_enc.size() / (32 * t);
  if (idx.size() != m) throw std::invalid_argument("eval_each_wire: one index per dealer");
  if (m == 0) return {};
  std::vector<int32_t> out(40 * m);
  std::vector<uint8_t> ok(m * t);
  group::edwards25519::detail::engine_must(kyb_pubpoly_eval_multi_enc_batch(commits_enc.data(), t, m, idx.data(), 1, nullptr, out.data(), ok.data()), "eval_each_wire");
  for (uint8_t f : ok) if (!f) throw MarshallingError("invalid Ed25519 curve point");
  std::vector<PubShare> r(m);
  for (size_t g = 0; g < m; ++g) { r[g].i = idx[g]; std::memcpy(r[g].v.ge, &out[40 * g], 160); }
  return r;
}
inline PubPoly sum_polys_wire(const std::vector<uint8_t>& commits_enc, size_t t, const std::optional<Point>& base) {
  wire_check(commits_enc, t);
  const size_t n = commits_enc.size() / (32 * t);
  if (n == 0) throw std::invalid_argument("sum_polys_wire: no polynomial");
  std::vector<int32_t> out(40 * t);
  std::vector<uint8_t> ok(n * t);
  group::edwards25519::detail::engine_must(kyb_sum_enc_batch(commits_enc.data(), t, n, 1, nullptr, out.data(), ok.data()), "sum_polys_wire");
  for (uint8_t f : ok) if (!f) throw MarshallingError("invalid Ed25519 curve point");
  PubPoly r;
  r.b = base;
  r.commits.resize(t);
  for (size_t j = 0; j < t; ++j) std::memcpy(r.commits[j].ge, &out[40 * j], 160);
  return r;
}

// poly.rs:534-563: the first t shares by index; x_i = i + 1
struct XYCommit { std::vector<size_t> idx; std::vector<Scalar> x; std::vector<Point> y; };
inline XYCommit xy_commit(const std::vector<std::optional<PubShare>>& shares, size_t t, size_t /*n*/) {
  std::vector<const PubShare*> sorted;
  for (const auto& s : shares) if (s) sorted.push_back(&*s);
  std::stable_sort(sorted.begin(), sorted.end(), [](const PubShare* a, const PubShare* b) { return a->i < b->i; });
  XYCommit r;
  for (const PubShare* s : sorted) {
    if (!r.idx.empty() && r.idx.back() == s->i) { r.y.back() = s->v; continue; }      // HashMap::insert: a repeated index overwrites
    r.idx.push_back(s->i);
    r.x.push_back(Scalar().set_int64((int64_t)(s->i + 1)));
    r.y.push_back(s->v);
    if (r.idx.size() == t) break;
  }
  return r;
}

namespace detail {
// Lagrange coefficients at 0 (poly.rs:585-594): prod_{j != i} x_j / (x_j - x_i)
inline std::vector<Scalar> lagrange_at_zero(const std::vector<Scalar>& x) {
  std::vector<Scalar> lam(x.size());
  for (size_t i = 0; i < x.size(); ++i) {
    Scalar num = Scalar().one(), den = Scalar().one();
    for (size_t j = 0; j < x.size(); ++j) {
      if (i == j) continue;
      num = num * x[j];
      den = den * Scalar().sub(x[j], x[i]);
    }
    lam[i] = Scalar().div(num, den);
  }
  return lam;
}
// The same coefficients from the engine (kyb_lagrange_coeffs_batch: one GPU lane per coefficient) for m share sets of t indices each —
// what recover_commit uses below; lagrange_at_zero above stays as the host restatement of the reference's loop (tests compare the two).
inline std::vector<Scalar> lagrange_at_zero_gpu(const std::vector<uint32_t>& indices, size_t m, size_t t) {
  std::vector<uint8_t> out(32 * m * t);
  group::edwards25519::detail::engine_must(kyb_lagrange_coeffs_batch(indices.data(), m, t, out.data()), "lagrange_at_zero");
  std::vector<Scalar> lam(m * t);
  for (size_t i = 0; i < m * t; ++i) std::memcpy(lam[i].v.data(), &out[32 * i], 32);
  return lam;
}
// scalars_public: Lagrange coefficients of public share indices — the engine may then use tables of the points (kyb_lincomb_public_batch)
inline std::vector<Point> lincomb(const std::vector<Scalar>& sc, const std::vector<Point>& pts, bool shared, size_t m, size_t t, const char* what,
                                  bool scalars_public = false) {
  std::vector<uint8_t> s(32 * m * t);
  std::vector<int32_t> p(40 * pts.size()), out(40 * m);
  for (size_t i = 0; i < m * t; ++i) std::memcpy(&s[32 * i], sc[i].v.data(), 32);
  for (size_t i = 0; i < pts.size(); ++i) std::memcpy(&p[40 * i], pts[i].limbs(), 160);
  const auto fn = scalars_public ? kyb_lincomb_public_batch : kyb_lincomb_batch;
  group::edwards25519::detail::engine_must(fn(s.data(), nullptr, p.data(), shared ? 1 : 0, m, t, nullptr, out.data(), nullptr), what);
  std::vector<Point> r(m);
  for (size_t g = 0; g < m; ++g) std::memcpy(r[g].ge, &out[40 * g], 160);
  return r;
}
}  // namespace detail

// poly.rs:566-603
inline Point recover_commit(const std::vector<std::optional<PubShare>>& shares, size_t t, size_t n) {
  XYCommit xy = xy_commit(shares, t, n);
  if (xy.x.size() < t) throw PolyError("not enough good public shares to reconstruct secret commitment");
  if (xy.x.empty()) return Point().null();
  // share indices are public: the coefficients come from the GPU (the reference's t^2 Scalar products run on one core) and the
  // combination may use point tables
  std::vector<uint32_t> idx(xy.idx.begin(), xy.idx.end());
  return detail::lincomb(detail::lagrange_at_zero_gpu(idx, 1, idx.size()), xy.y, false, 1, xy.x.size(), "recover_commit", true)[0];
}
// the same for many share sets in one launch (every peer's commitment of a DKG round)
inline std::vector<Point> recover_commit_batch(const std::vector<std::vector<std::optional<PubShare>>>& sets, size_t t, size_t n) {
  if (t == 0) return std::vector<Point>(sets.size(), Point().null());
  std::vector<uint32_t> idx;
  std::vector<Point> pts;
  for (const auto& shares : sets) {
    XYCommit xy = xy_commit(shares, t, n);
    if (xy.x.size() < t) throw PolyError("not enough good public shares to reconstruct secret commitment");
    idx.insert(idx.end(), xy.idx.begin(), xy.idx.end());
    pts.insert(pts.end(), xy.y.begin(), xy.y.end());
  }
  return detail::lincomb(detail::lagrange_at_zero_gpu(idx, sets.size(), t), pts, false, sets.size(), t, "recover_commit_batch", true);
}

// poly.rs:244-280 (recover_secret; xy_scalar :282-311 sorts by index, keeps the first t, a repeated index overwrites): the secret p(0) from
// t private shares.  The reference spends 2 t^2 Scalar products and t divisions on the Lagrange coefficients of the PUBLIC indices; those come
// from the engine (kyb_lagrange_coeffs_batch), the t products with the secret shares stay on the host.  Same canonical scalar.
inline Scalar recover_secret(const std::vector<std::optional<PriShare>>& shares, size_t t, size_t /*n*/) {
  std::vector<const PriShare*> sorted;
  for (const auto& s : shares) if (s) sorted.push_back(&*s);
  std::stable_sort(sorted.begin(), sorted.end(), [](const PriShare* a, const PriShare* b) { return a->i < b->i; });
  std::vector<uint32_t> idx;
  std::vector<Scalar> y;
  for (const PriShare* s : sorted) {
    if (!idx.empty() && idx.back() == (uint32_t)s->i) { y.back() = s->v; continue; }
    idx.push_back((uint32_t)s->i);
    y.push_back(s->v);
    if (idx.size() == t) break;
  }
  if (idx.size() < t) throw PolyError("not enough shares to recover secret");
  Scalar acc = Scalar().zero();
  if (idx.empty()) return acc;
  const std::vector<Scalar> lam = detail::lagrange_at_zero_gpu(idx, 1, idx.size());
  for (size_t i = 0; i < idx.size(); ++i) acc = acc + y[i] * lam[i];
  return acc;
}

// poly.rs:313-319 (minus_const: x - c) and :640-668 (lagrange_basis)
inline PriPoly minus_const(const Scalar& c) {
  PriPoly p;
  p.coeffs = {Scalar().neg(c), Scalar().one()};
  return p;
}
inline PriPoly lagrange_basis(size_t i, const std::vector<Scalar>& xs) {
  PriPoly basis;
  basis.coeffs = {Scalar().one()};
  Scalar acc = Scalar().one();
  for (size_t m = 0; m < xs.size(); ++m) {
    if (m == i) continue;
    basis = basis.mul(minus_const(xs[m]));
    Scalar den = Scalar().sub(xs[i], xs[m]);
    acc = acc * Scalar().inv(den);
  }
  for (Scalar& c : basis.coeffs) c = c * acc;
  return basis;
}
// poly.rs:607-634: sum_j L_j * y_j in point space = for every coefficient g: sum_j L_j[g] * y_j
inline PubPoly recover_pub_poly(const std::vector<std::optional<PubShare>>& shares, size_t t, size_t n) {
  XYCommit xy = xy_commit(shares, t, n);
  if (xy.x.size() < t) throw PolyError("not enough good public shares to reconstruct secret commitment");
  const size_t k = xy.x.size();
  PubPoly r;
  if (k == 0) return r;
  std::vector<Scalar> sc(k * k);
  for (size_t j = 0; j < k; ++j) {
    PriPoly basis = lagrange_basis(j, xy.x);
    for (size_t g = 0; g < k; ++g) sc[g * k + j] = basis.coeffs[g];
  }
  r.commits = detail::lincomb(sc, xy.y, true, k, k, "recover_pub_poly", true);      // Lagrange basis coefficients of public indices
  return r;
}

}  // namespace share
}  // namespace kyber
