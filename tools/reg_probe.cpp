// Experiment: can the page cache be handed to the copy engines directly?  mmap the BAM, hipHostRegister it piece by piece on T threads,
// hipMemcpyAsync each piece to the device, unregister.  Prints the aggregate GB/s and the share of the registration.
// usage: reg_probe <file> <threads> <piece MB> [flags: 0 default, 1 read-only]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int T = std::atoi(argv[2]);
    const size_t P = (size_t)std::atoi(argv[3]) << 20;
    const int ro = argc > 4 ? std::atoi(argv[4]) : 0;
    const int fd = open(argv[1], O_RDONLY);
    struct stat st; fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    uint8_t* m = (uint8_t*)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) { perror("mmap"); return 1; }
    uint8_t* dev = nullptr;
    if (hipMalloc(&dev, n + 4096) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
    for (int rep = 0; rep < 3; ++rep) {
        std::atomic<size_t> next{0};
        std::atomic<long long> reg_ns{0}, copy_ns{0};
        std::atomic<int> bad{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&]() {
            hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            for (;;) {
                const size_t off = next.fetch_add(P);
                if (off >= n) break;
                const size_t len = std::min(P, n - off);
                const auto a = std::chrono::steady_clock::now();
                hipError_t e = hipHostRegister(m + off, len, ro ? hipHostRegisterReadOnly : hipHostRegisterDefault);
                const auto b = std::chrono::steady_clock::now();
                if (e != hipSuccess) { if (!bad.fetch_add(1)) std::printf("hipHostRegister: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); break; }
                e = hipMemcpyAsync(dev + off, m + off, len, hipMemcpyHostToDevice, s);
                if (e == hipSuccess) e = hipStreamSynchronize(s);
                const auto c = std::chrono::steady_clock::now();
                if (e != hipSuccess) { if (!bad.fetch_add(1)) std::printf("copy: %s\n", hipGetErrorString(e)); }
                hipHostUnregister(m + off);
                reg_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count();
                copy_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(c - b).count();
            }
            hipStreamDestroy(s);
        });
        for (auto& x : th) x.join();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("rep %d: %d threads, %zu MB pieces, flags %d: %.3f s = %.1f GB/s (register %.0f ms, copy+wait %.0f ms summed over threads)%s\n", rep, T, P >> 20, ro, sec, n / sec * 1e-9,
                    reg_ns.load() * 1e-6, copy_ns.load() * 1e-6, bad.load() ? "  FAILED" : "");
        if (bad.load()) break;
    }
    return 0;
}
