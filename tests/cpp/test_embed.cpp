// Point::embed / Point::pick of the C++ mirror (host/edwards25519.hpp <- point.rs:90-92, 106-167) on the GPU engine over REPLAYED
// key streams: reads one vector per line on stdin — "<data hex | - for None | = for empty data> <stream hex>" — and prints
// "<encoding hex> <blocks consumed> <Point::data hex | - | !>" per line; tests/test_gpu_cpp_group.py compares with tests/golden/kats.json
// (known answers from the C oracle and the big-int model).
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../kyber-rs_amd/host/edwards25519.hpp"

using namespace kyber;
using namespace kyber::group::edwards25519;

struct ReplayStream : Stream {
  std::vector<uint8_t> bytes;
  size_t pos = 0;
  void xor_key_stream(uint8_t* dst, const uint8_t* src, size_t n) override {
    if (pos + n > bytes.size()) throw std::runtime_error("key stream exhausted");
    for (size_t i = 0; i < n; ++i) dst[i] = src[i] ^ bytes[pos + i];
    pos += n;
  }
};

static std::vector<uint8_t> unhex(const std::string& h) {
  std::vector<uint8_t> v(h.size() / 2);
  for (size_t i = 0; i < v.size(); ++i) v[i] = (uint8_t)std::stoi(h.substr(2 * i, 2), nullptr, 16);
  return v;
}
static std::string hex(const std::vector<uint8_t>& v) {
  static const char* d = "0123456789abcdef";
  std::string s;
  for (uint8_t b : v) { s.push_back(d[b >> 4]); s.push_back(d[b & 15]); }
  return s;
}

int main() {
  if (kyb_init(0) != KYB_OK) { std::printf("kyb_init failed: %s\n", kyb_last_error()); return 2; }
  std::string line;
  while (std::getline(std::cin, line)) {
    if (line.empty()) continue;
    std::istringstream is(line);
    std::string d, st;
    is >> d >> st;
    ReplayStream rand;
    rand.bytes = unhex(st);
    Point p;
    try {
      if (d == "-") p = Point().pick(rand);
      else {
        std::vector<uint8_t> data = d == "=" ? std::vector<uint8_t>() : unhex(d);
        static const uint8_t none_yet[1] = {0};
        p = Point().embed(data.empty() ? none_yet : data.data(), data.size(), rand);       // a non-null pointer even when empty: Some(&[])
      }
    } catch (const std::runtime_error& e) {
      std::printf("exhausted %zu -\n", rand.pos / 32);
      continue;
    }
    std::string out_data = "-";
    try { out_data = hex(p.data()); if (out_data.empty()) out_data = "="; } catch (const PointError&) { out_data = "!"; }
    std::printf("%s %zu %s\n", p.hex().c_str(), rand.pos / 32, out_data.c_str());
  }
  kyb_shutdown();
  return 0;
}
