// host_parallel.h -- the few lines of threading the host-side scene compile uses (BVH build, per-triangle records).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

namespace akr {

// Host threads for the build: AKR_HOST_THREADS, else the cgroup CPU quota (a GPU box shows 256 logical CPUs and grants 16 cores:
// 256 busy threads under that quota run slower than 16), else the hardware concurrency.
inline unsigned host_threads() {
    if (const char* e = std::getenv("AKR_HOST_THREADS")) {
        int v = std::atoi(e);
        if (v > 0) return (unsigned)std::min(v, 256);
    }
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        double per = 0.0;
        if (std::fscanf(f, "%63s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0.0) {
            const double cores = std::atof(q) / per;
            if (cores >= 1.0 && cores < (double)n) n = (unsigned)(cores + 0.5);
        }
        std::fclose(f);
    }
    return std::min(n, 64u);
}
// fn(chunk) for chunk in [0, n_chunks) on up to `threads` threads (the calling thread included)
template <typename F>
void parallel_chunks(unsigned n_chunks, unsigned threads, F&& fn) {
    if (n_chunks <= 1 || threads <= 1) {
        for (unsigned c = 0; c < n_chunks; c++) fn(c);
        return;
    }
    std::atomic<unsigned> next{0};
    std::atomic<bool> failed{false};
    std::exception_ptr first_error;  // an exception in a worker (std::bad_alloc in a subtree build ...) is re-thrown by the caller:
    std::mutex error_mutex;          // it must reach the C ABI's guard as a status code, not std::terminate the host process
    auto work = [&] {
        for (;;) {
            unsigned c = next.fetch_add(1);
            if (c >= n_chunks || failed.load()) break;
            try {
                fn(c);
            } catch (...) {
                std::lock_guard<std::mutex> lock(error_mutex);
                if (!first_error) first_error = std::current_exception();
                failed.store(true);
            }
        }
    };
    std::vector<std::thread> pool;
    const unsigned extra = std::min(threads, n_chunks) - 1;
    pool.reserve(extra);
    try {
        for (unsigned t = 0; t < extra; t++) pool.emplace_back(work);
    } catch (...) {  // the system refused another thread: the ones that started (and this one) do the work
    }
    work();
    for (auto& t : pool) t.join();
    if (first_error) std::rethrow_exception(first_error);
}


}  // namespace akr
