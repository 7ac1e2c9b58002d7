// cmx_hostpool.hpp -- a small persistent host thread pool for the memory-bound AoS -> SoA packing at set_packet /
// set_window.  Host only; no device code.
#pragma once
#include <stdint.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace cmx {

// split [0, n) over a few host threads (the AoS->SoA packing of millions of events is memory-bound on one core).
// The workers are created once per process and parked on a condition variable: spawning eight threads per call costs
// more than the packing they do (0.25 ms per call against ~0.1 ms of work for a 1M-event packet).
class HostPool {
 public:
  static HostPool &get() {
    static HostPool p;
    return p;
  }
  int workers() const { return (int)th_.size(); }
  // run job(k) for k in [0, parts) on the workers and the caller; returns when all are done.  One caller at a time
  // per process is enough here (packing is a fraction of a millisecond), so concurrent callers serialise.
  void run(int parts, const std::function<void(int)> &job) {
    std::lock_guard<std::mutex> serial(run_mutex_);
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &job;
      next_ = 0;
      parts_ = parts;
      pending_ = parts;
      generation_++;
    }
    cv_.notify_all();
    work();  // the caller takes parts too
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  HostPool() {
    unsigned hw = std::thread::hardware_concurrency();
    const int T = (int)(hw ? (hw > 8 ? 8 : hw) : 1);
    for (int t = 1; t < T; t++) th_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  void work() {
    for (;;) {
      int k;
      const std::function<void(int)> *job;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (!job_ || next_ >= parts_) return;
        k = next_++;
        job = job_;
      }
      (*job)(k);
      std::lock_guard<std::mutex> lk(m_);
      if (--pending_ == 0) done_.notify_all();
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      work();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_, run_mutex_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *job_ = nullptr;
  int next_ = 0, parts_ = 0, pending_ = 0;
  unsigned long long generation_ = 0;
  bool stop_ = false;
};

template <typename F>
void parallel_ranges(int64_t n, F fn, int64_t serial_below = 262144) {
  int T = HostPool::get().workers() + 1;
  if (n < serial_below) T = 1;
  if (T <= 1) { fn((int64_t)0, n); return; }
  const int64_t per = (n + T - 1) / T;
  const int parts = (int)((n + per - 1) / per);
  HostPool::get().run(parts, [&](int k) {
    const int64_t a = (int64_t)k * per, b = (a + per < n) ? a + per : n;
    if (a < b) fn(a, b);
  });
}

// argument checks only; the coordinate range is validated inside the packing pass (one sweep over the events instead

}  // namespace cmx
