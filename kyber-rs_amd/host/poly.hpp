// C++ mirror of the public-polynomial operations of kyber-rs that reach the engine.
//   PriPoly::commit   /root/reference src/share/poly.rs:195-206   t x mul(coeff, b)  -> one batch
//   PubPoly::eval     src/share/poly.rs:457-469                   Horner with x = i + 1
//   PubPoly::shares   src/share/poly.rs:472-478                   eval at 0..n-1 -> one batch
//   PubPoly::check    src/share/poly.rs:526-530                   eval(s.i) == mul(s.v, b)
#pragma once
#include <optional>
#include <vector>

#include "edwards25519.hpp"

namespace kyber {
namespace share {

using group::edwards25519::Point;
using group::edwards25519::Scalar;

struct PubShare { size_t i; Point v; };
struct PriShare { size_t i; Scalar v; };

class PubPoly {
 public:
  std::optional<Point> b;        // base point (None = the standard base)
  std::vector<Point> commits;

  size_t threshold() const { return commits.size(); }
  Point commit() const { return commits[0]; }

  std::vector<PubShare> eval_many(const std::vector<uint32_t>& idx) const {
    std::vector<int32_t> c(40 * commits.size()), out(40 * idx.size());
    for (size_t j = 0; j < commits.size(); ++j) std::memcpy(&c[40 * j], commits[j].ge, 160);
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
  // poly.rs:195-206
  PubPoly commit(const Point* base) const {
    PubPoly p;
    if (base) {
      p.b = *base;
      std::vector<Point> bs(coeffs.size(), *base);
      p.commits = Point::mul_batch(coeffs, &bs);
    } else {
      p.commits = Point::mul_batch(coeffs, nullptr);
    }
    return p;
  }
};

}  // namespace share
}  // namespace kyber
