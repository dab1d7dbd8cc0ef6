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
    std::printf("ok\n");
    return 0;
}
