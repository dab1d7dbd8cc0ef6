// Does the L2 keep what a wave has just written with byte stores?  Every wave owns a 64 KB region (as a resolve wave owns its block's output) and, round after
// round, writes the next 192 bytes of it and reads 8 bytes per lane from `back` bytes behind -- with byte stores (three per lane, as the literal tokens of
// k_lz_resolve3 are written), with one 4-byte store per lane (48 lanes), or with 16-byte stores (12 lanes: whole 64-byte lines).  Run under rocprofv3 --pmc
// TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum and --kernel-trace: requests that leave the L2 and time per variant.
//   hipcc --offload-arch=gfx950 -O3 -o build/l2_probe tools/l2_probe.cpp && build/l2_probe [waves] [rounds] [back]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
template <int MODE>
__global__ __launch_bounds__(64) void k_probe(uint8_t* buf, int rounds, int back, unsigned long long* sink) {
    uint8_t* out = buf + (size_t)blockIdx.x * 65536;
    const int lane = threadIdx.x;
    unsigned long long acc = 0;
    uint32_t pos = 4096;  // (the first reads look into bytes nobody wrote: the buffer is zeroed)
    for (int r = 0; r < rounds; ++r) {
        const uint32_t v = (uint32_t)(acc >> 7) + (uint32_t)r * 2654435761u + (uint32_t)lane;
        if (MODE == 0) { out[pos + 3 * lane] = (uint8_t)v; out[pos + 3 * lane + 1] = (uint8_t)(v >> 8); out[pos + 3 * lane + 2] = (uint8_t)(v >> 16); }
        else if (MODE == 1) { if (lane < 48) *(uint32_t*)(out + pos + 4 * lane) = v; }
        else { if (lane < 12) *(uint4*)(out + pos + 16 * lane) = make_uint4(v, v + 1, v + 2, v + 3); }
        unsigned long long w;
        __builtin_memcpy(&w, out + pos - back + 3 * lane, 8);  // the source of a match `back` bytes behind
        acc += w;
        pos += 192;
        if (pos + 192 + 16 > 65536) pos = 4096;
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}
int main(int argc, char** argv) {
    const int waves = argc > 1 ? std::atoi(argv[1]) : 8192, rounds = argc > 2 ? std::atoi(argv[2]) : 300, back = argc > 3 ? std::atoi(argv[3]) : 400;
    uint8_t* buf; unsigned long long* sink;
    CHK(hipMalloc((void**)&buf, (size_t)waves * 65536 + 4096)); CHK(hipMalloc((void**)&sink, 64));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            CHK(hipMemset(buf, 0, (size_t)waves * 65536 + 4096));
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(a, 0));
            if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(waves), dim3(64), 0, 0, buf, rounds, back, sink);
            else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(waves), dim3(64), 0, 0, buf, rounds, back, sink);
            else hipLaunchKernelGGL(k_probe<2>, dim3(waves), dim3(64), 0, 0, buf, rounds, back, sink);
            CHK(hipEventRecord(b, 0)); CHK(hipEventSynchronize(b));
            float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
            std::printf("mode %d (%s): %d waves x %d rounds, source %d bytes behind: %.3f ms, %.2f us per round\n", mode, mode == 0 ? "byte stores" : mode == 1 ? "4-byte stores" : "16-byte stores", waves, rounds, back, ms, 1e3 * ms / rounds * (waves > 8192 ? 8192.0 / waves : 1.0));
        }
    return 0;
}
