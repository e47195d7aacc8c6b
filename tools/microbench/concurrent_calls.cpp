// N host threads, each with its own context (kyb_ctx_create), issuing one-item host-pointer calls for a fixed time: calls per second
// of the whole process.  Build: g++ -O2 -std=c++17 -I include tools/microbench/concurrent_calls.cpp -o tools/_build/concurrent_calls -L kyber-rs_amd -lkyber_ed25519_hip -lpthread
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "kyber_ed25519.h"

int main(int argc, char** argv) {
  if (kyb_init(0) != KYB_OK) { printf("kyb_init failed: %s\n", kyb_last_error()); return 1; }
  uint8_t sc[32] = {7, 1, 2, 3}, enc[32];
  int32_t ext[40];
  if (kyb_mul_base_batch(sc, 1, enc, ext) != KYB_OK) return 1;
  printf("threads, op, calls_per_s, mean_call_us\n");
  for (int nt : {1, 4, 16, 64}) {
    for (int op = 0; op < 2; ++op) {
      std::atomic<long> total{0};
      std::atomic<bool> go{false}, stop{false};
      std::vector<std::thread> th;
      for (int i = 0; i < nt; ++i)
        th.emplace_back([&, i] {
          kyb_ctx* c = nullptr;
          if (kyb_ctx_create(0, 1, &c) != KYB_OK) { printf("ctx_create: %s\n", kyb_last_error()); return; }
          kyb_ctx_set_current(c);
          uint8_t s2[32]; memcpy(s2, sc, 32); s2[0] = (uint8_t)(i + 1);
          uint8_t o[32];
          auto call = [&] { return op == 0 ? kyb_mul_base_batch(s2, 1, o, nullptr) : kyb_mul_batch(s2, nullptr, ext, 1, o, nullptr, nullptr); };
          call();
          while (!go.load()) std::this_thread::yield();
          long n = 0;
          while (!stop.load()) { if (call() != KYB_OK) { printf("call failed: %s\n", kyb_last_error()); break; } ++n; }
          total += n;
          kyb_ctx_set_current(nullptr);
          kyb_ctx_destroy(c);
        });
      std::this_thread::sleep_for(std::chrono::milliseconds(300));
      auto t0 = std::chrono::steady_clock::now();
      go = true;
      std::this_thread::sleep_for(std::chrono::milliseconds(1500));
      stop = true;
      for (auto& t : th) t.join();
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      printf("%d, %s, %.0f, %.0f\n", nt, op == 0 ? "mul_base" : "mul", total.load() / dt, dt * nt / (double)total.load() * 1e6);
      fflush(stdout);
    }
  }
  return 0;
}
