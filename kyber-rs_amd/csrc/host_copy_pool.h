// Threaded memcpy for the host-pointer API (plain C++, no HIP): included by kernels.hip, and compiled on
// its own under ThreadSanitizer by tests/test_copy_pool.py.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace kyb {

// memcpy of large blocks on a few persistent host threads (one thread moves ~10 GB/s, PCIe takes 25+):
// the calling thread takes slices too and returns when every slice is done.
class CopyPool {
 public:
  struct Job { uint8_t* dst; const uint8_t* src; size_t bytes; };
  ~CopyPool() { stop(); }
  void stop() {
    { std::lock_guard<std::mutex> lk(m_); quit_ = true; ++gen_; }
    cv_.notify_all();
    for (std::thread& t : th_) t.join();
    th_.clear();
    quit_ = false;
  }
  // one caller at a time (the engine's host-pointer mutex is held)
  void run(const Job* jobs, int nj, int threads) {
    slices_.clear();
    for (int j = 0; j < nj; ++j)
      for (size_t o = 0; o < jobs[j].bytes; o += SLICE)
        slices_.push_back(Job{jobs[j].dst + o, jobs[j].src + o, jobs[j].bytes - o < SLICE ? jobs[j].bytes - o : SLICE});
    if (slices_.empty()) return;
    while ((int)th_.size() < threads - 1) {
      uint64_t now;
      { std::lock_guard<std::mutex> lk(m_); now = gen_; }
      th_.emplace_back([this, now] { worker(now); });           // sleeps until the next generation
    }
    next_.store(0);
    left_.store(slices_.size());
    { std::lock_guard<std::mutex> lk(m_); active_ = (int)th_.size(); ++gen_; }
    cv_.notify_all();
    drain();
    // every worker has to check in (even one that found nothing left) before slices_ may change again
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return active_ == 0 && left_.load() == 0; });
  }

 private:
  static constexpr size_t SLICE = (size_t)1 << 20;
  void drain() {
    for (;;) {
      const size_t k = next_.fetch_add(1);
      if (k >= slices_.size()) return;
      memcpy(slices_[k].dst, slices_[k].src, slices_[k].bytes);
      left_.fetch_sub(1);
    }
  }
  void worker(uint64_t seen) {
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (quit_) return;
      }
      drain();
      { std::lock_guard<std::mutex> lk(m_); --active_; }
      done_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::vector<Job> slices_;
  std::atomic<size_t> next_{0}, left_{0};
  std::mutex m_;
  std::condition_variable cv_, done_;
  uint64_t gen_ = 0;
  int active_ = 0;
  bool quit_ = false;
};

}  // namespace kyb
