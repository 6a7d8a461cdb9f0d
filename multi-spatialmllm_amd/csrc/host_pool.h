// A persistent pool of host worker threads for the ingest entry points (no device code here).
//
// mspa_read_depth_png_host / mspa_inflate_blocks_host / mspa_gather_blocks_host used to start their threads per call.  Starting
// a thread costs 50-100 us (stack mapping, TLS of the inflate tables): 64 threads for a 64-frame scene took longer to start than
// a frame takes to decode (tools/ingest_bench.py, round 5: 6.7 ms per scene on 64 threads against 1.56 ms per frame), and the
// loader keeps several scenes in flight, each with its own set.  The pool's threads are started once, on demand (never more
// than the process may use: affinity mask and cgroup CPU quota), and shared by all callers; a call hands the pool
// `n_threads - 1` helper tickets for its job and works on the job itself, so it never waits for a helper to start: if the pool
// is busy the caller simply does more of the items, and takes its unclaimed tickets back when it has run out of them.
//
// Not observable from outside: results, error strings and re-entrancy are those of the per-call threads.  The pool is leaked at
// exit (no destructor ordering against the interpreter's teardown) and rebuilt in a forked child (threads do not survive fork).
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <pthread.h>
#include <sched.h>
#include <thread>

namespace mspa {

class HostPool {
  public:
    static HostPool &get() {
        HostPool *p = instance().load(std::memory_order_acquire);
        if (!p) {
            static std::mutex make;
            std::lock_guard<std::mutex> g(make);
            p = instance().load(std::memory_order_acquire);
            if (!p) {
                p = new HostPool();
                static bool hooked = false;
                if (!hooked) {
                    pthread_atfork(nullptr, nullptr, [] { instance().store(nullptr, std::memory_order_release); });
                    hooked = true;
                }
                instance().store(p, std::memory_order_release);
            }
        }
        return *p;
    }

    // Run `fn` on up to n_threads threads at once, the caller among them; returns when every started copy has returned.
    // `fn` takes its work from a shared counter (the callers' pattern), so a copy that starts late finds nothing and leaves.
    void parallel(int n_threads, const std::function<void()> &fn) {
        if (n_threads <= 1) {
            fn();
            return;
        }
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->pending = n_threads - 1;
        {
            std::lock_guard<std::mutex> g(m_);
            for (int i = 0; i < n_threads - 1; ++i) queue_.push_back(job);
            const size_t want = std::min<size_t>(cap_, busy_ + queue_.size());
            while (n_workers_ < want) {
                try {
                    std::thread([this] { worker(); }).detach();
                    ++n_workers_;
                } catch (...) {
                    break;                                    // thread creation refused: the tickets wait for the workers there are
                }
            }
        }
        cv_.notify_all();
        fn();
        // The caller's share is done (the shared counter is exhausted): tickets nobody has claimed yet would only start copies
        // that find nothing -- and would make this call wait for a worker to get round to them behind other callers' real
        // work, or for ever if no worker thread could be started.  Take them back.
        int unclaimed = 0;
        {
            std::lock_guard<std::mutex> g(m_);
            for (auto it = queue_.begin(); it != queue_.end();) {
                if (it->get() == job.get()) {
                    it = queue_.erase(it);
                    ++unclaimed;
                } else {
                    ++it;
                }
            }
        }
        std::unique_lock<std::mutex> lk(job->m);
        job->pending -= unclaimed;
        job->cv.wait(lk, [&] { return job->pending == 0; });
    }

  private:
    struct Job {
        const std::function<void()> *fn = nullptr;
        int pending = 0;
        std::mutex m;
        std::condition_variable cv;
    };
    static std::atomic<HostPool *> &instance() {
        static std::atomic<HostPool *> p{nullptr};
        return p;
    }
    // The CPUs this process may really use: the affinity mask, cut down to the cgroup's CFS quota (cpu.max) -- a container on
    // a 256-thread host may hold 16 CPUs' worth of quota, and a pool that keeps more threads busy than that gets the WHOLE
    // group frozen for the rest of each 100 ms period (decode, staging and the thread feeding the GPU alike).  MSPA_HOST_CPUS
    // overrides.  mspa/hostinfo.py is the same rule on the Python side.
    static size_t effective_cpus() {
        if (const char *e = getenv("MSPA_HOST_CPUS")) {
            const long v = atol(e);
            if (v > 0) return (size_t)v;
        }
        size_t n = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) {
            const int c = CPU_COUNT(&set);
            if (c > 0) n = (size_t)c;
        }
        if (!n) n = 8;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            double period = 0;
            if (fscanf(f, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const double cpus = atof(q) / period;
                if (cpus >= 1 && (size_t)cpus < n) n = (size_t)cpus;
            }
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            double quota = 0, period = 0;
            const bool ok = fscanf(g, "%lf", &quota) == 1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (ok && fscanf(h, "%lf", &period) == 1 && quota > 0 && period > 0 && quota / period >= 1 &&
                    (size_t)(quota / period) < n)
                    n = (size_t)(quota / period);
                fclose(h);
            }
        }
        if (const char *e = getenv("LOCAL_WORLD_SIZE")) {       // the ranks of a node share its CPUs (torch.distributed.run)
            const long w = atol(e);
            if (w > 1) n = n / (size_t)w ? n / (size_t)w : 1;
        }
        return n;
    }
    HostPool() { cap_ = effective_cpus(); }
    void worker() {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !queue_.empty(); });
                job = std::move(queue_.front());
                queue_.pop_front();
                ++busy_;
            }
            try {
                (*job->fn)();
            } catch (...) {
            }
            {
                std::lock_guard<std::mutex> g(m_);
                --busy_;
            }
            {
                std::lock_guard<std::mutex> g(job->m);
                --job->pending;
            }
            job->cv.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Job>> queue_;
    size_t n_workers_ = 0, busy_ = 0, cap_ = 8;
};

}  // namespace mspa
