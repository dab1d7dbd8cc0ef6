// std::sort on several threads WITH THE SAME RESULT as libstdc++'s std::sort, tie order included.
//
// Why: the reference sorts with std::sort and comparators that leave ties (ReadRec.cpp:354,382; SegmentGraph.cpp:264), and the order
// of tied elements reaches the output (ledger B8); DESIGN.md section 0 pins "the permutation libstdc++'s introsort produces".  That
// permutation is a function of the comparisons alone, and libstdc++'s std::__sort is
//     __introsort_loop(first, last, 2 * lg(n))      median-of-three quicksort down to ranges of <= 16, heap sort below the depth limit
//     __final_insertion_sort(first, last)
// where every call of the loop only touches its own range [first, last): the right part of a partition is a recursive call, the left
// part the next turn of the loop.  Running those calls on different threads changes nothing about what any of them does -- same
// ranges, same depth limits, same comparisons -- so the array after the loop is the same, and the final insertion sort is run as it is.
// The pieces are libstdc++'s own (bits/stl_algo.h, GCC 11: __unguarded_partition_pivot, __introsort_loop, __partial_sort,
// __final_insertion_sort), called directly.  tests/test_host_logic.py compares with std::sort on tie-heavy inputs (sq_debug_parsort).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace sq {

template <class It, class Cmp>
void std_sort_parallel(It first, It last, Cmp comp, int threads) {
    const long n = (long)(last - first);
    if (threads <= 1 || n < (1 << 16)) { std::sort(first, last, comp); return; }
#if !defined(__GLIBCXX__)
    // The threaded form below is put together from libstdc++'s own introsort pieces; with another standard library there is
    // nothing to reproduce (its std::sort has tie orders of its own) and the plain call is all that can be said.
    std::sort(first, last, comp); return;
#else
    auto cmp = __gnu_cxx::__ops::__iter_comp_iter(comp);
    struct Task { It first, last; long depth; };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Task> queue;
    int busy = 0;  // tasks being worked on
    const long grain = std::max<long>(4096, n / (16L * threads));
    auto run = [&](Task t) {
        // == __introsort_loop(t.first, t.last, t.depth, cmp), with large right parts handed to the queue instead of recursed into
        while (t.last - t.first > 16) {
            if (t.depth == 0) { std::__partial_sort(t.first, t.last, t.last, cmp); return; }
            --t.depth;
            It cut = std::__unguarded_partition_pivot(t.first, t.last, cmp);
            if (t.last - cut > grain) {
                { std::lock_guard<std::mutex> lk(mu); queue.push_back(Task{cut, t.last, t.depth}); }
                cv.notify_one();
            } else std::__introsort_loop(cut, t.last, t.depth, cmp);
            t.last = cut;
        }
    };
    auto worker = [&]() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&]() { return !queue.empty() || busy == 0; });
            if (queue.empty()) { cv.notify_all(); return; }  // nothing queued and nobody working: done
            Task t = queue.back();
            queue.pop_back();
            ++busy;
            lk.unlock();
            run(t);
            lk.lock();
            --busy;
            if (busy == 0 && queue.empty()) cv.notify_all();
        }
    };
    queue.push_back(Task{first, last, (long)std::__lg(n) * 2});
    std::vector<std::thread> th;
    for (int i = 1; i < threads; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    std::__final_insertion_sort(first, last, cmp);
#endif
}

}  // namespace sq
