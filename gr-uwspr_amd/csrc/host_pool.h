// host_pool.h -- the persistent pool of host threads behind the sequential tail of the path
// (de-interleave + Fano per candidate: sync_and_demodulate_impl.cc:457-490 is a loop over independent
// candidates).  Threads are created once per process and sleep between jobs; a job is a parallel-for
// whose indices are handed out by one atomic counter, the calling thread works too.
#pragma once

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/uwspr_hip.h"

namespace uwspr {

int host_cpu_share();   // hardware threads capped by affinity and by the cgroup CPU quota, divided by the ranks sharing the host
int host_set_ranks(int ranks);   // uwspr_host_set_ranks

class host_pool {
 public:
  explicit host_pool(int nthreads);   // <= 0: host_cpu_share()
  ~host_pool();
  int size() const { return nworkers_ + 1; }
  // fn(i) for i in [0, n) on at most max_threads threads (<= 0: all), the caller among them; returns when all
  // are done.  Callers on different threads run their jobs side by side.
  void run(int n, int max_threads, const std::function<void(int)> &fn);
  static host_pool &shared();

 private:
  struct job {
    const std::function<void(int)> *fn = nullptr;
    int n = 0, chunk = 1, max_threads = 1;
    std::atomic<int> next{0};
    int threads = 0, done = 0;   // under m_
  };
  void worker();
  int drain(job &j);
  job *pick();
  int count_pending();
  std::vector<std::thread> threads_;
  int nworkers_ = 0;
  std::mutex m_;
  std::condition_variable cv_, cv_done_;
  std::vector<job *> jobs_;            // jobs in the pool (each lives on its caller's stack until it is done)
  size_t rr_ = 0;
  std::atomic<uint64_t> gen_a_{0};     // bumped per job: the workers' spin looks at it without the lock
  std::atomic<int> pending_{0};        // jobs with items still to hand out
  bool stop_ = false;
};

// one jiggered try of a candidate: the gates of cc:470 and, if it passes them, de-interleave + Fano.
// returns 1 decoded (message7 filled), 0 not decoded, -1 the try did not pass the gates (no Fano call)
int decode_try(const uwspr_demod_out *d, int idt, int8_t *message7);

// uwspr_decode_candidate from try `first` on
int decode_candidate_from(const uwspr_demod_out *d, int first, int8_t *message7, int32_t *idt_used,
                          int *fano_calls = nullptr);

}  // namespace uwspr
