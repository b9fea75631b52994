/*
 * rt_threads.hpp - host-side helpers of the scene set-up (flattening, tree builds, repacking): a worker pool that
 * lives as long as one set-up call, loops shared out over it, and stage stamps for timing the set-up.
 */
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

namespace rtown {

/* std::vector whose resize() leaves trivially constructible elements unwritten: the arrays of a large scene are filled
 * by several threads right after they are sized, and writing zeros first (one thread, every page touched) would cost as
 * much as the fill. */
template <typename T>
struct NoInit : std::allocator<T> {
    template <typename U> struct rebind { using other = NoInit<U>; };
    NoInit() = default;
    template <typename U> NoInit(const NoInit<U> &) {}
    template <typename U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new ((void *)p) U; }
    template <typename U, typename... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
template <typename T> using raw_vector = std::vector<T, NoInit<T>>;

/* Stage stamps on stderr when RT_UPLOAD_TIMING is set in the environment: the time since this thread's previous stamp
 * (nullptr: only restart the clock). */
inline void stage_tick(const char *what) {
    static const bool on = getenv("RT_UPLOAD_TIMING") != nullptr;
    if (!on) return;
    static thread_local std::chrono::steady_clock::time_point last;
    const auto now = std::chrono::steady_clock::now();
    if (what) fprintf(stderr, "%-48s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
}

/* worker threads for the host-side builds: OMP_NUM_THREADS when it is set (job schedulers size it to the process's share
 * of the machine), else the hardware's count, at most 64 */
inline int host_threads() {
    int t = 0;
    if (const char *e = getenv("OMP_NUM_THREADS")) t = atoi(e);
    if (t <= 0) t = (int)std::thread::hardware_concurrency();
    if (t <= 0) t = 1;
    return t > 64 ? 64 : t;
}

/* Worker threads that live as long as one set-up call and sleep between its parallel passes: the top of a build makes
 * a hundred short passes, and starting fifteen threads for each would cost more than the passes do. A thread that
 * wants one declares a PoolScope; parallel_parts and parallel_tasks called by that thread then use it, and start
 * threads of their own otherwise (another thread's call, or a call from inside a pass). */
class Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)> *fn = nullptr;
    uint64_t gen = 0;
    int parts = 0, pending = 0;
    bool stop = false;
    std::exception_ptr first_error; /* of the current pass: an exception on any thread (std::bad_alloc in a builder pass) is
                                       carried to the thread that called run() instead of ending the process */

    void loop(int p) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            if (p >= parts) continue;
            const std::function<void(int)> *f = fn;
            lk.unlock();
            std::exception_ptr err;
            try { (*f)(p); } catch (...) { err = std::current_exception(); }
            lk.lock();
            if (err && !first_error) first_error = err;
            if (--pending == 0) cv_done.notify_one();
        }
    }

public:
    bool busy = false; /* inside run(): a pass that itself asks for threads gets fresh ones */
    explicit Pool(int threads) {
        for (int p = 1; p < threads; p++) th.emplace_back([this, p] { loop(p); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : th) t.join();
    }
    int size() const { return (int)th.size() + 1; }
    /* f(0) .. f(n - 1), one thread each, n <= size(); the caller's thread takes 0. An exception thrown by any f(p) is
     * rethrown here once every thread has finished its part: the workers hold references into the caller's stack, so the
     * pass is always waited for, and the pool is left ready for the next pass. */
    void run(int n, const std::function<void(int)> &f) {
        {
            std::lock_guard<std::mutex> lk(m);
            fn = &f; parts = n; pending = n - 1; gen++;
            first_error = nullptr;
        }
        busy = true;
        cv_work.notify_all();
        std::exception_ptr mine;
        try { f(0); } catch (...) { mine = std::current_exception(); }
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
        fn = nullptr;
        busy = false;
        std::exception_ptr err = mine ? mine : first_error;
        first_error = nullptr;
        lk.unlock();
        if (err) std::rethrow_exception(err);
    }
};

inline Pool *&this_threads_pool() {
    static thread_local Pool *p = nullptr;
    return p;
}

struct PoolScope {
    Pool *mine = nullptr;
    explicit PoolScope(int threads) {
        if (!this_threads_pool() && threads > 1) this_threads_pool() = mine = new Pool(threads);
    }
    ~PoolScope() {
        if (mine) { this_threads_pool() = nullptr; delete mine; }
    }
    PoolScope(const PoolScope &) = delete;
    PoolScope &operator=(const PoolScope &) = delete;
};

/* f(0) .. f(n - 1) on n threads (the caller's is one of them) */
inline void on_threads(int n, const std::function<void(int)> &f) {
    if (n <= 1) { f(0); return; }
    Pool *pool = this_threads_pool();
    if (pool && !pool->busy && n <= pool->size()) { pool->run(n, f); return; }
    /* fresh threads: the same rule - every thread is joined before the first exception goes on to the caller */
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err((size_t)n);
    auto guarded = [&f, &err](int p) { try { f(p); } catch (...) { err[(size_t)p] = std::current_exception(); } };
    try {
        for (int p = 1; p < n; p++) th.emplace_back(guarded, p);
    } catch (...) { /* a thread could not be started: the parts without one are run here */
        for (int p = (int)th.size() + 1; p < n; p++) guarded(p);
    }
    guarded(0);
    for (auto &t : th) t.join();
    for (const auto &e : err)
        if (e) std::rethrow_exception(e);
}

/* fn(part, begin, end) on `parts` contiguous pieces of [0, n), one thread each */
template <typename F>
inline void parallel_parts(int64_t n, int parts, F fn) {
    if (parts <= 1 || n < 2) { fn(0, (int64_t)0, n); return; }
    on_threads(parts, [&](int p) { fn(p, n * p / parts, n * (p + 1) / parts); });
}

/* fn(task) for every task in [0, n), handed out one at a time to `threads` threads */
template <typename F>
inline void parallel_tasks(size_t n, int threads, F fn) {
    std::atomic<size_t> next(0);
    on_threads((int)std::min<size_t>((size_t)(threads < 1 ? 1 : threads), n ? n : 1), [&](int) {
        for (size_t q; (q = next.fetch_add(1)) < n;) fn(q);
    });
}

} // namespace rtown
