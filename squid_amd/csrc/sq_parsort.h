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
// The final insertion sort, one thread's work over the whole array (a tenth of a second for ten million elements), is split as well when
// the caller says its comparator is a STRICT WEAK ORDER (`strict_weak`): a partition leaves every element of its left part <= pivot <=
// every element of its right part, so no element of the right part is ever less than one of the left part, and the linear insert of the
// final pass -- which moves an element left past the elements GREATER than it -- never carries anything across a cut of the loop.  The
// array between two cuts is therefore sorted by the final pass on its own: __final_insertion_sort on every stretch between the cuts at
// which the loop handed work to the queue (its guarded first sixteen never look in front of the stretch; the elements behind them stop
// inside their own leaf of at most sixteen) gives what the one pass over everything gives.  A comparator that is not a strict weak order
// (the reference's FrontSmallerThan, ReadRec.cpp:382) offers no such guarantee and keeps the single pass.
// The pieces are libstdc++'s own (bits/stl_algo.h, GCC 11: __unguarded_partition_pivot, __introsort_loop, __partial_sort,
// __final_insertion_sort), called directly.  tests/test_host_logic.py compares with std::sort on tie-heavy inputs (sq_debug_parsort).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace sq {

template <class It, class Cmp>
void std_sort_parallel(It first, It last, Cmp comp, int threads, bool strict_weak = false) {
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
    std::vector<It> cuts;  // where the loop handed its right part to the queue
    int busy = 0;  // tasks being worked on
    const long grain = std::max<long>(4096, n / (16L * threads));
    auto run = [&](Task t) {
        // == __introsort_loop(t.first, t.last, t.depth, cmp), with large right parts handed to the queue instead of recursed into
        while (t.last - t.first > 16) {
            if (t.depth == 0) { std::__partial_sort(t.first, t.last, t.last, cmp); return; }
            --t.depth;
            It cut = std::__unguarded_partition_pivot(t.first, t.last, cmp);
            if (t.last - cut > grain) {
                { std::lock_guard<std::mutex> lk(mu); queue.push_back(Task{cut, t.last, t.depth}); cuts.push_back(cut); }
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
    if (!strict_weak || cuts.empty()) { std::__final_insertion_sort(first, last, cmp); return; }
    cuts.push_back(first); cuts.push_back(last);
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    std::atomic<size_t> next{0};
    auto finish = [&]() { for (size_t k; (k = next.fetch_add(1)) + 1 < cuts.size();) std::__final_insertion_sort(cuts[k], cuts[k + 1], cmp); };
    th.clear();
    for (int i = 1; i < threads; ++i) th.emplace_back(finish);
    finish();
    for (auto& t : th) t.join();
#endif
}

}  // namespace sq
