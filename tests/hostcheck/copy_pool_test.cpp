// Hammer kyb::CopyPool (kyber-rs_amd/csrc/host_copy_pool.h): many runs of varying shapes and thread
// counts, results compared byte for byte.  Built with -fsanitize=thread by tests/test_copy_pool.py.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../kyber-rs_amd/csrc/host_copy_pool.h"

int main() {
  std::mt19937_64 rng(7);
  const size_t cap = (size_t)5 << 20;
  std::vector<uint8_t> src(cap), dst(cap);
  for (size_t i = 0; i < cap; ++i) src[i] = (uint8_t)(rng() >> 56);
  kyb::CopyPool pool;
  for (int round = 0; round < 60; ++round) {
    const int threads = 1 + (int)(rng() % 6);
    const int nj = (int)(rng() % 4);                     // 0..3 jobs, may be empty
    kyb::CopyPool::Job jobs[4];
    size_t off = 0;
    std::fill(dst.begin(), dst.end(), 0);
    for (int j = 0; j < nj; ++j) {
      size_t bytes = (rng() % 5 == 0) ? 0 : (size_t)(rng() % ((size_t)5 << 19));
      if (off + bytes > cap) bytes = cap - off;
      jobs[j] = kyb::CopyPool::Job{dst.data() + off, src.data() + off, bytes};
      off += bytes;
    }
    pool.run(jobs, nj, threads);
    for (size_t i = 0; i < cap; ++i) {
      const uint8_t want = i < off ? src[i] : 0;
      if (dst[i] != want) { std::printf("FAILED round %d byte %zu\n", round, i); return 1; }
    }
    if (round == 30) pool.stop();                       // restartable after stop()
  }
  std::printf("OK\n");
  return 0;
}
