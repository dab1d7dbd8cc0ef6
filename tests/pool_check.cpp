// HostPool::parallel_for from a task of the pool itself, on pools of 0, 1 and 3 threads: must return (tests/test_abi.py)
#include "../squid_amd/csrc/sq_internal.h"
#include <cstdio>
int main() {
    for (int threads : {0, 1, 3}) {
        sq::HostPool pool(threads);
        std::atomic<long> sum{0};
        auto loop = [&]() { pool.parallel_for(1000, 1 << 20, [&](int i) { sum += i; }); };
        loop();
        if (threads) {
            // as the cluster table runs: a task of the pool that starts loops (and a loop inside a loop)
            auto fut = pool.submit([&]() { loop(); pool.parallel_for(4, 4, [&](int) { loop(); }); return 1; });
            if (fut.get() != 1) return 1;
        }
        const long want = 499500L * (threads ? 6 : 1);
        if (sum.load() != want) { std::printf("threads %d: sum %ld, expected %ld\n", threads, sum.load(), want); return 1; }
    }
    // an exception thrown by a body -- on the calling thread or on a helper -- reaches the caller of parallel_for, after every index has
    // been accounted for (no helper still inside the body), and the pool goes on working
    for (int threads : {0, 1, 3}) {
        sq::HostPool pool(threads);
        for (int bad : {0, 500, 999}) {
            std::atomic<int> ran{0}, inside{0};
            bool caught = false;
            try { pool.parallel_for(1000, 1 << 20, [&](int i) { ++inside; ++ran; if (i == bad) { --inside; throw std::bad_alloc(); } --inside; }); }
            catch (const std::bad_alloc&) { caught = true; }
            if (!caught || inside.load() != 0 || ran.load() < 1) { std::printf("threads %d: exception of index %d lost (caught %d, inside %d)\n", threads, bad, (int)caught, inside.load()); return 1; }
        }
        std::atomic<long> sum{0};
        pool.parallel_for(100, 8, [&](int i) { sum += i; });
        if (sum.load() != 4950) { std::printf("threads %d: pool unusable after an exception\n", threads); return 1; }
    }
    std::printf("ok\n");
    return 0;
}
