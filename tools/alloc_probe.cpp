// Probe (GPU box): what does device memory cost to get?  hipMalloc measured 40-60 ms per GiB on the MI355X boxes of this pool.
// build: hipcc -O2 -std=c++17 -o build/alloc_probe tools/alloc_probe.cpp -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define TRY(x) ((x) == hipSuccess ? true : (std::printf("  (%s failed: %s)\n", #x, hipGetErrorString(hipGetLastError())), false))

__global__ void k_touch(uint32_t* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i; }

int main() {
    auto t0 = clk::now();
    CHK(hipSetDevice(0)); CHK(hipFree(nullptr));
    std::printf("runtime up: %.1f ms\n", ms(t0));
    const size_t G = (size_t)1 << 30;
    {   // one large vs many small
        auto a = clk::now(); void* p; CHK(hipMalloc(&p, 4 * G)); std::printf("hipMalloc 4 GiB: %.1f ms\n", ms(a)); CHK(hipFree(p));
        a = clk::now(); std::vector<void*> v(64); for (auto& q : v) CHK(hipMalloc(&q, 64 << 20)); std::printf("64 x hipMalloc 64 MiB: %.1f ms\n", ms(a)); for (auto q : v) CHK(hipFree(q));
        a = clk::now(); std::vector<void*> w(2048); for (auto& q : w) CHK(hipMalloc(&q, 2 << 20)); std::printf("2048 x hipMalloc 2 MiB: %.1f ms\n", ms(a)); for (auto q : w) CHK(hipFree(q));
    }
    {   // four threads at once
        auto a = clk::now();
        std::vector<std::thread> th; std::vector<void*> ps(4);
        for (int t = 0; t < 4; ++t) th.emplace_back([&, t]() { (void)hipSetDevice(0); (void)hipMalloc(&ps[t], G); });
        for (auto& x : th) x.join();
        std::printf("4 threads x hipMalloc 1 GiB: %.1f ms\n", ms(a));
        for (auto p : ps) CHK(hipFree(p));
    }
    {   // stream-ordered allocator
        hipStream_t s; CHK(hipStreamCreate(&s));
        void* p = nullptr;
        auto a = clk::now();
        if (TRY(hipMallocAsync(&p, 4 * G, s))) { CHK(hipStreamSynchronize(s)); std::printf("hipMallocAsync 4 GiB: %.1f ms\n", ms(a));
            a = clk::now(); CHK(hipFreeAsync(p, s)); CHK(hipStreamSynchronize(s)); std::printf("hipFreeAsync: %.1f ms\n", ms(a));
            a = clk::now(); if (TRY(hipMallocAsync(&p, 4 * G, s))) { CHK(hipStreamSynchronize(s)); std::printf("hipMallocAsync 4 GiB again (pool): %.1f ms\n", ms(a)); CHK(hipFreeAsync(p, s)); CHK(hipStreamSynchronize(s)); } }
    }
    {   // virtual memory management: reserve an address range, map physical chunks as they are needed
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0;
        if (TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum))) {
            std::printf("VMM granularity %zu\n", gran);
            void* va = nullptr;
            auto a = clk::now();
            if (TRY(hipMemAddressReserve(&va, 8 * G, 0, nullptr, 0))) {
                std::printf("hipMemAddressReserve 8 GiB: %.2f ms\n", ms(a));
                const size_t chunk = (size_t)256 << 20;
                hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
                std::vector<hipMemGenericAllocationHandle_t> hs;
                a = clk::now();
                bool ok = true;
                for (int i = 0; i < 16 && ok; ++i) {
                    hipMemGenericAllocationHandle_t h;
                    ok = TRY(hipMemCreate(&h, chunk, &prop, 0)) && TRY(hipMemMap((char*)va + i * chunk, chunk, 0, h, 0)) && TRY(hipMemSetAccess((char*)va + i * chunk, chunk, &acc, 1));
                    if (ok) hs.push_back(h);
                }
                std::printf("16 x (hipMemCreate + hipMemMap + hipMemSetAccess) of 256 MiB = 4 GiB: %.1f ms\n", ms(a));
                if (ok) { a = clk::now(); hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, nullptr, (uint32_t*)va, 4 * G / 4); CHK(hipDeviceSynchronize()); std::printf("kernel writes the 4 GiB: %.2f ms\n", ms(a)); }
                for (size_t i = 0; i < hs.size(); ++i) { (void)hipMemUnmap((char*)va + i * chunk, chunk); (void)hipMemRelease(hs[i]); }
                (void)hipMemAddressFree(va, 8 * G);
            }
        }
    }
    {   // does a kernel keep running while another thread allocates?
        uint32_t* p; CHK(hipMalloc((void**)&p, G));
        hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, s, p, G / 4); CHK(hipStreamSynchronize(s));
        CHK(hipEventRecord(e0, s));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, s, p, G / 4);
        CHK(hipEventRecord(e1, s));
        auto a = clk::now();
        void* q; CHK(hipMalloc(&q, 4 * G));
        const double m = ms(a);
        CHK(hipStreamSynchronize(s));
        float k = 0; CHK(hipEventElapsedTime(&k, e0, e1));
        std::printf("hipMalloc 4 GiB beside 200 running kernels: %.1f ms; the kernels took %.1f ms (alone: see next line)\n", m, k);
        CHK(hipEventRecord(e0, s));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, s, p, G / 4);
        CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
        CHK(hipEventElapsedTime(&k, e0, e1));
        std::printf("200 kernels alone: %.1f ms\n", k);
        // launching from this thread while another allocates
        std::thread th([&]() { (void)hipSetDevice(0); void* r; auto b = clk::now(); (void)hipMalloc(&r, 4 * G); std::printf("  (background hipMalloc 4 GiB: %.1f ms)\n", ms(b)); });
        a = clk::now();
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, s, p, 1 << 16);
        const double l = ms(a);
        CHK(hipStreamSynchronize(s));
        std::printf("200 small launches issued in %.2f ms, done after %.2f ms while another thread allocates\n", l, ms(a));
        th.join();
    }
    return 0;
}
