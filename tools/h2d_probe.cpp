// Probe (GPU box): how fast do the bytes of a file in the page cache reach HBM?
//   h2d_probe <file> : T threads pread() 8 MiB pieces into page-locked buffers of their own and send them off asynchronously;
//   also: the BGZF header walk with one pread per header, hipMalloc / hipFree of large buffers, hipHostMalloc.
// build: hipcc -O2 -std=c++17 -o build/h2d_probe tools/h2d_probe.cpp -lpthread
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// keeps every CU busy with a workgroup that holds 96 KB of LDS and spins on it (the shape of the token pass), for about `spin` rounds
__global__ void k_busy(unsigned long long spin, unsigned* sink) {
    extern __shared__ unsigned lds[];
    unsigned x = threadIdx.x;
    for (unsigned long long i = 0; i < spin; ++i) { lds[(x * 17 + i) & 16383] = x; x = x * 1664525u + lds[(x >> 7) & 16383]; }
    if (x == 0x12345678u) *sink = x;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    auto t0 = clk::now();
    CHK(hipSetDevice(0));
    CHK(hipFree(nullptr));
    std::printf("HIP runtime + context: %.1f ms\n", ms(t0));
    int fd = open(argv[1], O_RDONLY);
    struct stat st;
    fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    for (size_t gb : {1, 4, 4, 8}) {
        void* p = nullptr;
        auto a = clk::now();
        CHK(hipMalloc(&p, gb << 30));
        const double m = ms(a);
        a = clk::now();
        CHK(hipMemsetAsync(p, 0, gb << 30, nullptr)); CHK(hipDeviceSynchronize());
        const double m2 = ms(a);
        a = clk::now();
        CHK(hipFree(p));
        std::printf("hipMalloc %zu GiB: %.2f ms, first memset %.2f ms, hipFree %.2f ms\n", gb, m, m2, ms(a));
    }
    {
        void* p = nullptr;
        auto a = clk::now();
        CHK(hipHostMalloc(&p, (size_t)256 << 20, hipHostMallocDefault));
        std::printf("hipHostMalloc 256 MiB: %.2f ms\n", ms(a));
        CHK(hipHostFree(p));
    }
    uint8_t* dst = nullptr;
    CHK(hipMalloc((void**)&dst, n + 4096));
    unsigned* sink = nullptr;
    CHK(hipMalloc((void**)&sink, 4));
    CHK(hipFuncSetAttribute((const void*)k_busy, hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10));
    hipStream_t busy_stream;
    CHK(hipStreamCreateWithFlags(&busy_stream, hipStreamNonBlocking));
    for (int mode = 0; mode < 4; ++mode) {  // 0: pread, 1: memcpy from a fresh mapping (faults included), 2: pread beside a kernel that fills every CU, 3: the same with 32 MiB pieces
        const size_t P = (size_t)(mode == 3 ? 32 : 8) << 20;
        for (int T : {4, 8, 16, 32}) {
            if (mode >= 2 && T != 16) continue;
            if (mode >= 2) hipLaunchKernelGGL(k_busy, dim3(256), dim3(192), 96 << 10, busy_stream, 3000000ull, sink);
            const uint8_t* map = nullptr;
            if (mode == 1) map = (const uint8_t*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            std::vector<uint8_t*> pin((size_t)2 * T);
            std::vector<hipStream_t> sq((size_t)T);
            std::vector<hipEvent_t> ev((size_t)2 * T);
            auto a0 = clk::now();
            for (auto& p : pin) CHK(hipHostMalloc((void**)&p, P, hipHostMallocDefault));
            for (auto& s : sq) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            for (auto& e : ev) CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            const double setup = ms(a0);
            std::atomic<size_t> next{0};
            auto a = clk::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t]() {
                    (void)hipSetDevice(0);
                    bool used[2] = {false, false};
                    int b = 0;
                    for (;;) {
                        const size_t off = next.fetch_add(P);
                        if (off >= n) break;
                        const size_t len = std::min(P, n - off);
                        if (used[b]) (void)hipEventSynchronize(ev[(size_t)2 * t + b]);
                        if (mode != 1) { size_t got = 0; while (got < len) { ssize_t r = pread(fd, pin[(size_t)2 * t + b] + got, len - got, (off_t)(off + got)); if (r <= 0) break; got += (size_t)r; } }
                        else std::memcpy(pin[(size_t)2 * t + b], map + off, len);
                        (void)hipMemcpyAsync(dst + off, pin[(size_t)2 * t + b], len, hipMemcpyHostToDevice, sq[(size_t)t]);
                        (void)hipEventRecord(ev[(size_t)2 * t + b], sq[(size_t)t]);
                        used[b] = true; b ^= 1;
                    }
                    (void)hipStreamSynchronize(sq[(size_t)t]);
                });
            for (auto& x : th) x.join();
            const double w = ms(a);
            const char* what[4] = {"pread", "memcpy from fresh mmap", "pread beside a busy kernel", "pread beside a busy kernel, 32 MiB pieces"};
            std::printf("%s, %2d threads: %.1f ms = %.1f GB/s (setup of buffers/streams/events %.1f ms)\n", what[mode], T, w, (double)n / w / 1e6, setup);
            if (mode >= 2) { auto b0 = clk::now(); CHK(hipStreamSynchronize(busy_stream)); std::printf("  (the busy kernel ran %.1f ms longer)\n", ms(b0)); }
            for (auto& p : pin) CHK(hipHostFree(p));
            for (auto& s : sq) CHK(hipStreamDestroy(s));
            for (auto& e : ev) CHK(hipEventDestroy(e));
            if (map) munmap((void*)map, n);
        }
    }
    {   // BGZF header walk, one pread per header (isize of the block before + header of this one)
        auto a = clk::now();
        size_t p = 0, blocks = 0;
        uint8_t h[64];
        while (p + 18 <= n) {
            if (pread(fd, h, 32, (off_t)p) < 18 || h[0] != 0x1f || h[1] != 0x8b) break;
            const unsigned xlen = h[10] | (h[11] << 8);
            if (xlen != 6 || h[12] != 'B' || h[13] != 'C') break;
            p += (size_t)(h[16] | (h[17] << 8)) + 1;
            ++blocks;
        }
        std::printf("header walk with pread: %zu blocks in %.1f ms\n", blocks, ms(a));
    }
    return 0;
}
