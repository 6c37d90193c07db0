// host_parallel.h -- the host side's one threading primitive: contiguous ranges of an index space on a few std::threads.
//
// The client-side work (fractional encoding of 10^5..10^6 weights, one ChaCha20 keystream + three transforms per encrypted pixel) is independent per item and
// bit-identical however it is split, so it runs on up to CRC_HOST_THREADS threads (default: the hardware's, at most 16 -- a GPU box's CPU share -- divided by
// the
// ranks of the node when there are several: LOCAL_WORLD_SIZE).  Threads come from
// ONE budget per library: a caller that is itself one of many threads (benchkit's client encrypts images on a pool) finds the budget spent and runs its range
// inline, so nested parallelism never multiplies thread counts.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>

namespace crc_host {
inline int thread_limit()
{
    static const int lim = [] {
        int v = 0;
        if (const char *e = std::getenv("CRC_HOST_THREADS")) v = std::atoi(e);
        if (v <= 0) {
            const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
            v = std::min(16, hw);
            // one process per GPU: the ranks of a node share its cores (LOCAL_WORLD_SIZE is torch.distributed.run's, CRC_LOCAL_WORLD bench_host's own) -- eight
            // ranks encoding weights at once must not start 8 x 16 threads
            int ranks = 1;
            for (const char *name : {"CRC_LOCAL_WORLD", "LOCAL_WORLD_SIZE"})
                if (const char *e = std::getenv(name)) { const int r = std::atoi(e); if (r > 1) { ranks = r; break; } }
            if (ranks > 1) v = std::max(1, std::min(v, hw / ranks));
        }
        return std::min(v, 64);
    }();
    return lim;
}
inline std::atomic<int> &budget() { static std::atomic<int> b{thread_limit() - 1}; return b; }    // threads beside the caller's own

// fn(begin, end) over [0, count) in contiguous ranges of at least `grain` items; returns when all ranges are done.  fn must not throw.
template <class F> void parallel_for(size_t count, size_t grain, F &&fn)
{
    if (count == 0) return;
    size_t want = std::min<size_t>((count + grain - 1) / std::max<size_t>(1, grain), (size_t)thread_limit());
    int extra = 0;
    if (want > 1) {
        int have = budget().load(std::memory_order_relaxed);
        while (have > 0) {
            const int take = std::min<int>(have, (int)want - 1);
            if (budget().compare_exchange_weak(have, have - take, std::memory_order_acq_rel)) { extra = take; break; }
        }
    }
    if (extra == 0) { fn((size_t)0, count); return; }
    // the threads taken go back to the budget however this scope is left
    struct Return { int n; ~Return() { budget().fetch_add(n, std::memory_order_acq_rel); } } give_back{extra};
    const size_t parts = (size_t)extra + 1, per = (count + parts - 1) / parts;
    std::vector<std::thread> th;
    th.reserve(extra);
    size_t inline_from = count;                 // ranges from here on run on the caller's thread: a thread that could not be started (EAGAIN under a process or
    for (size_t p = 1; p < parts; p++) {        // thread limit) must not unwind through the joinable ones -- nothing throws across the C ABI above this
        const size_t b = std::min(count, p * per), e = std::min(count, b + per);
        if (b >= e) continue;
        try { th.emplace_back([&fn, b, e] { fn(b, e); }); }
        catch (...) { inline_from = b; break; }
    }
    fn((size_t)0, std::min(count, per));
    if (inline_from < count) fn(inline_from, count);
    for (auto &t : th) t.join();
}
}  // namespace crc_host
