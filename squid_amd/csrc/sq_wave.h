// The handful of wave-level operations the speculative token pass (sq_inflate_spec.inc) is written in, with two backends:
//   * gfx950 (default): DPP / readlane / ballot, LDS through address-space-3 pointers;
//   * SQ_WAVE_EMU: 64 coroutines in one host thread (ucontext), every wave-wide operation a rendezvous -- the kernel source itself runs
//     on the CPU, so its logic (the fixpoint over the 64 stretches, the table builder, the corner cases of RFC 1951) is debugged and
//     fuzzed against zlib without a GPU (tools/inflate_emu.cpp).  The emulator is test infrastructure; the product never builds it.
// Every wave-wide operation must be reached by all 64 lanes (uniform control flow); the emulator asserts that.
#pragma once
#include <cstdint>

#ifdef SQ_WAVE_EMU
#include <ucontext.h>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define WV_FN static inline
namespace wv {
typedef uint32_t lds_u32;
typedef uint8_t lds_u8;
struct Emu {
    ucontext_t sched, ctx[64];
    int cur = 0;
    bool finished[64];
    int tag[64];
    unsigned long long x[64];
    long n_rendezvous = 0;
};
inline Emu*& emu() { static Emu* e = nullptr; return e; }
WV_FN int lane() { return emu()->cur; }
inline void rendezvous(int tag) {  // every lane stops here; the scheduler checks that all of them came from the same place
    Emu* e = emu();
    e->tag[e->cur] = tag;
    swapcontext(&e->ctx[e->cur], &e->sched);
}
WV_FN void sync() { rendezvous(1); }
WV_FN unsigned long long ballot(bool p) {
    Emu* e = emu();
    e->x[e->cur] = p ? 1 : 0;
    rendezvous(2);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) m |= e->x[i] << i;
    rendezvous(3);
    return m;
}
WV_FN bool any(bool p) { return ballot(p) != 0; }
WV_FN uint32_t shfl(uint32_t v, int src) {  // (src may differ per lane)
    Emu* e = emu();
    e->x[e->cur] = v;
    rendezvous(4);
    const uint32_t r = (uint32_t)e->x[src & 63];
    rendezvous(5);
    return r;
}
WV_FN uint32_t shfl_up1(uint32_t v, uint32_t lane0) { const uint32_t r = shfl(v, lane() - 1); return lane() == 0 ? lane0 : r; }
WV_FN uint32_t bcast(uint32_t v, int src) { return shfl(v, src); }  // src uniform
WV_FN uint32_t scan_incl_add(uint32_t v) {
    Emu* e = emu();
    e->x[e->cur] = v;
    rendezvous(6);
    uint32_t r = 0;
    for (int i = 0; i <= e->cur; ++i) r += (uint32_t)e->x[i];
    rendezvous(7);
    return r;
}
WV_FN unsigned long long lanemask_lt() { return lane() ? (~0ull >> (64 - lane())) : 0ull; }
WV_FN uint32_t uniform(uint32_t v) { return v; }
WV_FN int popc64(unsigned long long m) { return __builtin_popcountll(m); }
WV_FN int ctz64(unsigned long long m) { return m ? __builtin_ctzll(m) : 64; }
WV_FN int clz64(unsigned long long m) { return m ? __builtin_clzll(m) : 64; }
WV_FN uint32_t brev32(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; ++i) if (v >> i & 1) r |= 1u << (31 - i); return r; }
WV_FN uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { sh &= 31; return sh ? (lo >> sh) | (hi << (32 - sh)) : lo; }
WV_FN uint32_t bfe(uint32_t v, uint32_t off, uint32_t width) { return width ? (v >> off) & (0xffffffffu >> (32 - width)) : 0u; }
WV_FN uint32_t lds_atomic_max(lds_u32* p, uint32_t v) { const uint32_t o = *p; if (v > o) *p = v; return o; }
WV_FN uint32_t lds_atomic_add(lds_u32* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
struct u32x4 { uint32_t x, y, z, w; };
WV_FN u32x4 load16(const uint32_t* p) { return u32x4{p[0], p[1], p[2], p[3]}; }
// (LDS accesses of any alignment)
WV_FN void lds_st32(lds_u8* p, uint32_t v) { std::memcpy(p, &v, 4); }
WV_FN void lds_st64(lds_u8* p, unsigned long long v) { std::memcpy(p, &v, 8); }
WV_FN uint32_t lds_ld32(const lds_u8* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
// run fn(arg) as a wave of 64 lanes; returns when every lane has returned
inline void run_wave(void (*fn)(void*), void* arg, size_t stack_bytes = 256 << 10) {
    Emu e;
    emu() = &e;
    static char* stacks = nullptr;
    static size_t stacks_each = 0;
    if (!stacks || stacks_each != stack_bytes) { std::free(stacks); stacks = (char*)std::malloc(64 * stack_bytes); stacks_each = stack_bytes; }
    struct Tramp { static void go(unsigned lo, unsigned hi, unsigned alo, unsigned ahi) {
        void (*f)(void*) = (void (*)(void*))(((uintptr_t)hi << 32) | lo);
        f((void*)(((uintptr_t)ahi << 32) | alo));
        Emu* em = emu(); em->finished[em->cur] = true; swapcontext(&em->ctx[em->cur], &em->sched); } };
    for (int i = 0; i < 64; ++i) {
        e.finished[i] = false; e.tag[i] = 0;
        getcontext(&e.ctx[i]);
        e.ctx[i].uc_stack.ss_sp = stacks + (size_t)i * stack_bytes; e.ctx[i].uc_stack.ss_size = stack_bytes; e.ctx[i].uc_link = nullptr;
        makecontext(&e.ctx[i], (void (*)())Tramp::go, 4, (unsigned)((uintptr_t)fn & 0xffffffffu), (unsigned)((uintptr_t)fn >> 32), (unsigned)((uintptr_t)arg & 0xffffffffu), (unsigned)((uintptr_t)arg >> 32));
    }
    for (;;) {
        int nfin = 0;
        for (int i = 0; i < 64; ++i) {  // (lanes in descending order every other round: a value read before the lane that writes it has run shows up)
            const int l = (e.n_rendezvous & 1) ? 63 - i : i;
            e.cur = l;
            if (!e.finished[l]) swapcontext(&e.sched, &e.ctx[l]);
            if (e.finished[l]) ++nfin;
        }
        ++e.n_rendezvous;
        if (nfin == 64) break;
        if (nfin != 0) { std::fprintf(stderr, "wave emulator: %d lanes returned while others wait at a wave-wide operation\n", nfin); std::abort(); }
        for (int i = 1; i < 64; ++i) if (e.tag[i] != e.tag[0]) { std::fprintf(stderr, "wave emulator: lanes 0 and %d wait at different operations (%d, %d)\n", i, e.tag[0], e.tag[i]); std::abort(); }
    }
    emu() = nullptr;
}
}  // namespace wv

#else  // ------------------------------------------------------------------------------------------------ gfx950
#define WV_FN static __device__ __forceinline__
namespace wv {
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
WV_FN int lane() { return (int)(threadIdx.x & 63u); }
WV_FN void sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
WV_FN unsigned long long ballot(bool p) { return __ballot(p); }
WV_FN bool any(bool p) { return __any(p); }
WV_FN uint32_t shfl(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v); }
WV_FN uint32_t shfl_up1(uint32_t v, uint32_t lane0) { return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
WV_FN uint32_t bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src)); }
WV_FN uint32_t scan_incl_add(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return x;
}
WV_FN unsigned long long lanemask_lt() { return ~0ull >> 1 >> (63 - lane()); }
WV_FN uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
WV_FN int popc64(unsigned long long m) { return __popcll(m); }
WV_FN int ctz64(unsigned long long m) { return m ? __builtin_ctzll(m) : 64; }
WV_FN int clz64(unsigned long long m) { return m ? __builtin_clzll(m) : 64; }
WV_FN uint32_t brev32(uint32_t v) { return __brev(v); }
WV_FN uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
WV_FN uint32_t bfe(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(v, off, width); }
WV_FN uint32_t lds_atomic_max(lds_u32* p, uint32_t v) { return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_FN uint32_t lds_atomic_add(lds_u32* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
typedef uint4 u32x4;
WV_FN u32x4 load16(const uint32_t* p) { return *(const uint4*)p; }
// (LDS accesses of any alignment: gfx950 takes them)
WV_FN void lds_st32(lds_u8* p, uint32_t v) { typedef uint32_t __attribute__((aligned(1))) u; *(__attribute__((address_space(3))) u*)p = v; }
WV_FN void lds_st64(lds_u8* p, unsigned long long v) { typedef unsigned long long __attribute__((aligned(1))) u; *(__attribute__((address_space(3))) u*)p = v; }
WV_FN uint32_t lds_ld32(const lds_u8* p) { typedef uint32_t __attribute__((aligned(1))) u; return *(const __attribute__((address_space(3))) u*)p; }
}  // namespace wv
#endif
