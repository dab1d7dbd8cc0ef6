// Checks sq::std_sort_parallel (squid_amd/csrc/sq_parsort.h) against std::sort: same permutation, tie order included, also with a
// comparator that is not a strict weak order (the reference's FrontSmallerThan is not one).  Built and run by tests/test_abi.py.
#include "../squid_amd/csrc/sq_parsort.h"
#include <cstdio>
#include <random>
#include <chrono>
struct E { int key; int id; long pad[2]; };
int main() {
    for (int trial = 0; trial < 6; ++trial) {
        size_t n = trial < 3 ? 1000000 : 200000 + trial * 1777;
        int range = trial == 0 ? 1000 : trial == 1 ? 1000000 : trial == 2 ? 3 : 50000;
        std::mt19937 rng(trial);
        std::vector<E> a(n);
        for (size_t i = 0; i < n; ++i) a[i] = E{(int)(rng() % range), (int)i, {0, 0}};
        if (trial == 4) std::sort(a.begin(), a.end(), [](const E& x, const E& y) { return x.key < y.key; });  // presorted
        if (trial == 5) for (size_t i = 0; i < n; ++i) a[i].key = (int)(n - i) / 3;  // descending with ties (deep recursion)
        std::vector<E> b = a;
        auto cmp = [](const E& x, const E& y) { return x.key < y.key; };
        auto t0 = std::chrono::steady_clock::now();
        std::sort(a.begin(), a.end(), cmp);
        auto t1 = std::chrono::steady_clock::now();
        sq::std_sort_parallel(b.begin(), b.end(), cmp, 8);
        auto t2 = std::chrono::steady_clock::now();
        bool same = true;
        for (size_t i = 0; i < n; ++i) if (a[i].id != b[i].id) { same = false; break; }
        if (!same) { std::printf("trial %d differs\n", trial); return 1; }
        {   // a strict weak order: the final insertion pass split at the cuts of the loop
            std::vector<E> c2(n);
            { std::mt19937 r2(trial); for (size_t i = 0; i < n; ++i) c2[i] = E{(int)(r2() % range), (int)i, {0, 0}}; }
            if (trial == 4) std::sort(c2.begin(), c2.end(), [](const E& x, const E& y) { return x.key < y.key; });
            if (trial == 5) for (size_t i = 0; i < n; ++i) c2[i].key = (int)(n - i) / 3;
            auto t3 = std::chrono::steady_clock::now();
            sq::std_sort_parallel(c2.begin(), c2.end(), cmp, 8, true);
            auto t4 = std::chrono::steady_clock::now();
            for (size_t i = 0; i < n; ++i) if (a[i].id != c2[i].id) { std::printf("trial %d differs with the split final pass at %zu\n", trial, i); return 1; }
            std::printf("        split final pass %.3f s\n", std::chrono::duration<double>(t4 - t3).count());
        }
        std::printf("trial %d n %zu same %d  std %.3f s par %.3f s\n", trial, n, (int)same, std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
    }
    {   // a comparator without transitivity of equivalence: elements compare by `key` only when both are "typed" the same way
        const size_t n = 400000;
        std::mt19937 rng(99);
        std::vector<E> a(n);
        for (size_t i = 0; i < n; ++i) a[i] = E{(int)(rng() % 5000), (int)i, {(long)(rng() % 3), 0}};
        std::vector<E> b = a;
        auto cmp = [](const E& x, const E& y) { return (x.pad[0] == 0 || y.pad[0] == 0) ? false : x.key < y.key; };
        std::sort(a.begin(), a.end(), cmp);
        sq::std_sort_parallel(b.begin(), b.end(), cmp, 6);
        for (size_t i = 0; i < n; ++i) if (a[i].id != b[i].id) { std::printf("non-strict comparator differs\n"); return 1; }
    }
    std::printf("ok\n");
    return 0;
}
