// The two helper threads of the per-call input path (engine_inputs.hip): the conversion worker pool and the
// launcher thread.  Included at the end of engine.h: drs_engine owns one of each through unique_ptr.
#pragma once

namespace drs {
namespace eng {

// ---- host-side worker pool for the per-call input pass ----------------------------------------
// drs_forward_inputs converts 160 k indices per RMC1 query on the host; one thread doing that
// (plus the Python call) capped the PCIe-inclusive path at 12 k queries/s.  The
// tables of a query are independent, so they are spread over a few workers.  Workers spin for a
// short while after a job (the next query usually follows within microseconds) and then sleep on
// a condition variable; the calling thread always takes part, so a pool of zero workers is just
// the plain loop.
class HostPool {
 public:
  explicit HostPool(int workers) {
    for (int i = 0; i < workers; ++i) th_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_.store(true, std::memory_order_release);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int workers() const { return (int)th_.size(); }
  // fn(i) for i in [0, n), on the caller and the workers; returns when all are done
  void run(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (th_.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    // a worker of the PREVIOUS job may still be between its last pending_ decrement and its next
    // next_ increment inside work(): re-arming fn_ / n_ / next_ under it would be a data race, and a
    // worker that grabs an item before pending_ is stored would leave run() spinning forever
    // (ADVICE r2).  Wait until nobody is inside work(), then arm pending_ BEFORE next_.
    while (active_.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
    fn_ = &fn; n_.store(n, std::memory_order_relaxed);
    pending_.store(n, std::memory_order_relaxed);
    next_.store(0, std::memory_order_release);
    {
      std::lock_guard<std::mutex> l(mu_);     // pairs with the sleepers' predicate check
      gen_.fetch_add(1, std::memory_order_release);
    }
    if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
    work();
    while (pending_.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
    next_.store(1 << 30, std::memory_order_release);   // closed: a worker that wakes up late finds no item
  }

 private:
  void work() {
    active_.fetch_add(1, std::memory_order_acq_rel);
    for (;;) {
      const int i = next_.fetch_add(1, std::memory_order_acq_rel);
      if (i >= n_.load(std::memory_order_relaxed)) break;
      (*fn_)(i);
      pending_.fetch_sub(1, std::memory_order_acq_rel);
    }
    active_.fetch_sub(1, std::memory_order_acq_rel);
  }
  void loop() {
    // gen_ is 0 when the constructor starts the workers: a worker that gets its first time slice only
    // after run() -- or the destructor -- has already bumped gen_ must still notice that bump (reading
    // gen_ here instead left such a worker asleep for ever and the destructor's join() with it: pools
    // that are created and destroyed without work in between, small models on a busy host)
    uint64_t seen = 0;
    for (;;) {
      // spin ~50 us for the next job, then sleep
      bool got = false;
      for (int spin = 0; spin < 20000; ++spin) {
        if (gen_.load(std::memory_order_acquire) != seen) { got = true; break; }
        __builtin_ia32_pause();
      }
      if (!got) {
        std::unique_lock<std::mutex> l(mu_);
        sleepers_.fetch_add(1, std::memory_order_acq_rel);
        cv_.wait(l, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        sleepers_.fetch_sub(1, std::memory_order_acq_rel);
      }
      seen = gen_.load(std::memory_order_acquire);
      if (stop_.load(std::memory_order_acquire)) return;
      work();
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0};
  std::atomic<int> next_{1 << 30}, pending_{0}, sleepers_{0}, active_{0};   // (next_ past any n_ while idle)
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> n_{0};
  std::atomic<bool> stop_{false};
};


// The launcher thread of the per-call input path: takes jobs in FIFO order and makes their HIP calls
// (finish_inputs).  Spins ~50 us for the next job, then sleeps.
class Launcher {
 public:
  explicit Launcher(drs_engine* e) : e_(e), th_([this] { loop(); }) {}
  ~Launcher() {
    { std::lock_guard<std::mutex> l(mu_); stop_ = true; }
    cv_.notify_all();
    th_.join();
  }
  void push(int slot, int mode, int32_t bs, size_t used, bool need_off) {
    e_->launch_state[slot].store(1, std::memory_order_release);
    { std::lock_guard<std::mutex> l(mu_); q_.push_back(Job{slot, mode, bs, used, need_off}); }
    pushed_.fetch_add(1, std::memory_order_release);
    if (sleeping_.load(std::memory_order_acquire)) cv_.notify_one();
  }
  // every job handed over so far has been launched (other entry points call this before they touch
  // streams or slots themselves)
  void drain() {
    while (done_.load(std::memory_order_acquire) != pushed_.load(std::memory_order_acquire)) __builtin_ia32_pause();
  }

 private:
  struct Job { int slot; int mode; int32_t bs; size_t used; bool need_off; };
  bool pop(Job* j) {
    std::lock_guard<std::mutex> l(mu_);
    if (q_.empty()) return false;
    *j = q_.front();
    q_.erase(q_.begin());
    return true;
  }
  void loop() {
    (void)hipSetDevice(e_->device);
    for (;;) {
      Job j;
      bool got = false;
      for (int spin = 0; spin < 20000 && !got; ++spin) {
        if (done_.load(std::memory_order_relaxed) != pushed_.load(std::memory_order_acquire)) got = pop(&j);
        else __builtin_ia32_pause();
      }
      if (!got) {
        std::unique_lock<std::mutex> l(mu_);
        sleeping_.store(true, std::memory_order_release);
        cv_.wait(l, [&] { return stop_ || !q_.empty(); });
        sleeping_.store(false, std::memory_order_release);
        if (q_.empty()) { if (stop_) return; continue; }
        j = q_.front();
        q_.erase(q_.begin());
      }
      Slot& s = e_->slots[j.slot];
      s.launch_rc = finish_inputs(e_, s, j.mode, j.bs, j.used, j.need_off);
      if (s.launch_rc) { std::lock_guard<std::mutex> l(e_->err_mu); s.launch_err = e_->err; }
      e_->launch_state[j.slot].store(2, std::memory_order_release);
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  drs_engine* e_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Job> q_;
  std::atomic<uint64_t> pushed_{0}, done_{0};
  std::atomic<bool> sleeping_{false};
  bool stop_ = false;
  std::thread th_;      // (last: the members above exist before it starts)
};

}  // namespace eng
}  // namespace drs
