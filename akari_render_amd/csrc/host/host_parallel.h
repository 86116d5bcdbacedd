// host_parallel.h -- the few lines of threading the host-side scene compile uses (BVH build, per-triangle records).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    auto work = [&] {
        for (;;) {
            unsigned c = next.fetch_add(1);
            if (c >= n_chunks) break;
            fn(c);
        }
    };
    std::vector<std::thread> pool;
    const unsigned extra = std::min(threads, n_chunks) - 1;
    pool.reserve(extra);
    for (unsigned t = 0; t < extra; t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
}


}  // namespace akr
