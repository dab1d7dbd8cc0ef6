// The exchange of a chromosome-sharded run, carried out by the library itself (sq_exchange(ctx, comm); 8(e):
// "one RCCL all-gather over xGMI").  sq_build_graph / sq_call_sv return SQ_NEED_EXCHANGE with this rank's payload in c->xbuf;
// sq_exchange all-gathers the payloads of all ranks over the transport installed on the context and hands them to the library --
// no host language in the loop.  Transports: RCCL (sq_rccl_init: ncclAllGather on device buffers, rank = sq_params.rank) or any
// fixed-size all-gather the caller supplies (sq_set_allgather: MPI, a gloo shim in the tests).
//
// Payloads differ in length from rank to rank and from exchange to exchange, an all-gather wants equal pieces: every rank sends one
// piece of SQ_X_PIECE bytes = [total length | the first bytes of its payload]; nearly all exchanges fit (stream boundaries, seed
// nodes, breakpoint counts are a few hundred bytes) and are ONE collective.  Only when some rank's payload is longer -- the data
// exchange of a large graph -- a second all-gather carries the remainders, padded to the longest.
#include <cstring>
#include <dlfcn.h>

#include <rccl/rccl.h>  // (types and prototypes only: the library itself is bound at run time, below)

#include "sq_internal.h"

namespace sq {

constexpr int64_t SQ_X_PIECE = 16384;
constexpr int64_t SQ_X_MAX_PAYLOAD = (int64_t)1 << 34;  // no rank's payload is anywhere near (node sums + reduced edges: megabytes); beyond = a corrupt length

// RCCL is bound when a sharded run first asks for it (dlopen): librccl.so carries half a gigabyte of device code for every
// architecture, and a single-GPU `squid` process that links it pays for mapping and registering all of it at start-up.
struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    RcclApi() {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        CommCount = (decltype(CommCount))dlsym(h, "ncclCommCount");
        CommUserRank = (decltype(CommUserRank))dlsym(h, "ncclCommUserRank");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        ok = GetUniqueId && CommInitRank && CommDestroy && AllGather && CommCount && CommUserRank && GetErrorString;
    }
};
static const RcclApi& rccl() { static RcclApi api; return api; }

struct RcclTransport {
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    hipStream_t stream = nullptr;
    uint8_t *d_send = nullptr, *d_recv = nullptr, *h_pin = nullptr;
    size_t cap_send = 0, cap_recv = 0, cap_pin = 0;
    int device = 0;
    ~RcclTransport() {
        if (d_send) (void)hipFree(d_send);
        if (d_recv) (void)hipFree(d_recv);
        if (h_pin) (void)hipHostFree(h_pin);
        if (stream) (void)hipStreamDestroy(stream);
        if (comm && own_comm) (void)rccl().CommDestroy(comm);
    }
    int allgather(const void* send, int64_t nbytes, void* recv, int world) {
        if (hipSetDevice(device) != hipSuccess) return SQ_E_HIP;
        const size_t ns = (size_t)nbytes, nr = (size_t)nbytes * (size_t)world;
        if (ns > cap_send) { if (d_send) (void)hipFree(d_send); d_send = nullptr; cap_send = 0; if (hipMalloc((void**)&d_send, ns + ns / 2) != hipSuccess) return SQ_E_HIP; cap_send = ns + ns / 2; }
        if (nr > cap_recv) { if (d_recv) (void)hipFree(d_recv); d_recv = nullptr; cap_recv = 0; if (hipMalloc((void**)&d_recv, nr + nr / 2) != hipSuccess) return SQ_E_HIP; cap_recv = nr + nr / 2; }
        if (hipMemcpyAsync(d_send, send, ns, hipMemcpyHostToDevice, stream) != hipSuccess) return SQ_E_HIP;
        if (rccl().AllGather(d_send, d_recv, ns, ncclUint8, comm, stream) != ncclSuccess) return SQ_E_HIP;
        if (hipMemcpyAsync(recv, d_recv, nr, hipMemcpyDeviceToHost, stream) != hipSuccess) return SQ_E_HIP;
        if (hipStreamSynchronize(stream) != hipSuccess) return SQ_E_HIP;
        return SQ_OK;
    }
};

static int rccl_trampoline(void* user, const void* send, int64_t nbytes, void* recv) {
    sq_ctx* c = (sq_ctx*)user;
    return c->rccl ? c->rccl->allgather(send, nbytes, recv, c->P.world_size) : SQ_E_ARG;
}

void exchange_release(sq_ctx* c) { c->rccl.reset(); }

}  // namespace sq

using namespace sq;

extern "C" {

int sq_set_allgather(sq_ctx* c, sq_allgather_fn fn, void* user) {
    if (!c) return SQ_E_ARG;
    c->x_allgather = fn; c->x_user = user;
    return SQ_OK;
}

int sq_rccl_unique_id(void* id128) {
    if (!id128) return SQ_E_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId travels as 128 bytes");
    ncclUniqueId id;
    if (!rccl().ok || rccl().GetUniqueId(&id) != ncclSuccess) return SQ_E_HIP;
    std::memcpy(id128, &id, sizeof id);
    return SQ_OK;
}

static int rccl_install(sq_ctx* c, ncclComm_t comm, bool own) {
    std::shared_ptr<RcclTransport> t = std::make_shared<RcclTransport>();
    t->comm = comm; t->own_comm = own; t->device = c->P.device;
    if (hipSetDevice(c->P.device) != hipSuccess || hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) return fail(c, SQ_E_HIP, "cannot create the exchange stream");
    c->rccl = t;
    c->x_allgather = rccl_trampoline; c->x_user = c;
    return SQ_OK;
}

int sq_rccl_available(void) { return rccl().ok ? 1 : 0; }

int sq_rccl_release(sq_ctx* c) {
    if (!c) return SQ_E_ARG;
    if (c->rccl) { c->rccl.reset(); if (c->x_allgather == rccl_trampoline) { c->x_allgather = nullptr; c->x_user = nullptr; } }
    return SQ_OK;
}

// The whole RCCL path of sq_exchange on ONE device: dlopen, ncclGetUniqueId, ncclCommInitRank with a world of one, the transport's
// all-gather (host piece -> device -> ncclAllGather -> host) for the fixed 16 KiB piece and for a 1 MiB remainder, bytes compared.
int sq_debug_rccl_selftest(sq_ctx* c) {
    if (!c) return SQ_E_ARG;
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, SQ_E_HIP, "hipSetDevice");
    if (!rccl().ok) return fail(c, SQ_E_HIP, "librccl.so.1 cannot be loaded");
    ncclUniqueId id;
    ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) return fail(c, SQ_E_HIP, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r));
    ncclComm_t comm = nullptr;
    r = rccl().CommInitRank(&comm, 1, id, 0);
    if (r != ncclSuccess) return fail(c, SQ_E_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    RcclTransport t;
    t.comm = comm; t.own_comm = true; t.device = c->P.device;
    if (hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking) != hipSuccess) return fail(c, SQ_E_HIP, "cannot create the exchange stream");
    int n = 0, me = -1;
    if (rccl().CommCount(comm, &n) != ncclSuccess || rccl().CommUserRank(comm, &me) != ncclSuccess || n != 1 || me != 0) return fail(c, SQ_E_HIP, "self-test communicator is not {1 rank, rank 0}");
    for (const int64_t nbytes : {SQ_X_PIECE, (int64_t)1 << 20}) {
        std::vector<uint8_t> send((size_t)nbytes), recv((size_t)nbytes, 0);
        uint64_t x = 0x9E3779B97F4A7C15ull ^ (uint64_t)nbytes;
        for (auto& b : send) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; b = (uint8_t)x; }
        const int rc = t.allgather(send.data(), nbytes, recv.data(), 1);
        if (rc) return fail(c, rc, "self-test all-gather failed");
        if (send != recv) return fail(c, SQ_E_HIP, "self-test all-gather returned different bytes");
    }
    return SQ_OK;
}

int sq_rccl_init(sq_ctx* c, const void* id128) {
    if (!c || !id128) return SQ_E_ARG;
    if (c->P.world_size <= 1) return fail(c, SQ_E_ARG, "sq_rccl_init needs sq_params.world_size > 1");
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, SQ_E_HIP, "hipSetDevice");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    if (!rccl().ok) return fail(c, SQ_E_HIP, "librccl.so.1 cannot be loaded");
    ncclComm_t comm = nullptr;
    const ncclResult_t r = rccl().CommInitRank(&comm, c->P.world_size, id, c->P.rank);
    if (r != ncclSuccess) return fail(c, SQ_E_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    return rccl_install(c, comm, true);
}

int sq_rccl_attach(sq_ctx* c, void* nccl_comm) {
    if (!c || !nccl_comm) return SQ_E_ARG;
    int n = 0, r = -1;
    if (!rccl().ok) return fail(c, SQ_E_HIP, "librccl.so.1 cannot be loaded");
    if (rccl().CommCount((ncclComm_t)nccl_comm, &n) != ncclSuccess || rccl().CommUserRank((ncclComm_t)nccl_comm, &r) != ncclSuccess) return fail(c, SQ_E_ARG, "not a communicator");
    if (n != c->P.world_size || r != c->P.rank) return fail(c, SQ_E_ARG, "communicator size / rank differ from sq_params.world_size / rank");
    return rccl_install(c, (ncclComm_t)nccl_comm, false);
}

static int exchange_body(sq_ctx* c);
int sq_exchange(sq_ctx* c) {
    if (!c) return SQ_E_ARG;
    try { return exchange_body(c); }
    catch (const std::bad_alloc&) { return fail(c, SQ_E_CAPACITY, "sharded run: out of host memory during an exchange"); }
    catch (const std::exception& e) { return fail(c, SQ_E_ARG, std::string("sharded run: ") + e.what()); }
}
static int exchange_body(sq_ctx* c) {
    if (!c->x_pending) return fail(c, SQ_E_ARG, "no exchange is pending (sq_build_graph / sq_call_sv return SQ_NEED_EXCHANGE first)");
    if (!c->x_allgather) return fail(c, SQ_E_ARG, "no transport installed (sq_rccl_init / sq_rccl_attach / sq_set_allgather)");
    const int W = c->P.world_size;
    const int64_t mine = (int64_t)c->xbuf.size(), head = SQ_X_PIECE - 8;
    std::vector<uint8_t> piece((size_t)SQ_X_PIECE, 0), got((size_t)SQ_X_PIECE * (size_t)W);
    std::memcpy(piece.data(), &mine, 8);
    if (mine) std::memcpy(piece.data() + 8, c->xbuf.data(), (size_t)std::min(mine, head));
    int rc = c->x_allgather(c->x_user, piece.data(), SQ_X_PIECE, got.data());
    if (rc) return fail(c, rc < 0 ? rc : SQ_E_HIP, "all-gather failed");
    c->x_collectives++;
    std::vector<int64_t> len((size_t)W);
    int64_t longest = 0;
    for (int r = 0; r < W; ++r) {
        std::memcpy(&len[(size_t)r], got.data() + (size_t)r * SQ_X_PIECE, 8);
        if (len[(size_t)r] < 0 || len[(size_t)r] > SQ_X_MAX_PAYLOAD) return fail(c, SQ_E_ARG, "sharded run: malformed exchange piece (a peer is out of step)");
        longest = std::max(longest, len[(size_t)r]);
    }
    std::vector<uint8_t> rest_got;
    const int64_t rest = longest > head ? longest - head : 0;
    if (rest) {  // the remainders, padded to the longest
        std::vector<uint8_t> rs((size_t)rest, 0);
        if (mine > head) std::memcpy(rs.data(), c->xbuf.data() + head, (size_t)(mine - head));
        rest_got.resize((size_t)rest * (size_t)W);
        rc = c->x_allgather(c->x_user, rs.data(), rest, rest_got.data());
        if (rc) return fail(c, rc < 0 ? rc : SQ_E_HIP, "all-gather failed");
        c->x_collectives++;
    }
    std::vector<uint8_t> all;
    for (int r = 0; r < W; ++r) {
        const int64_t n = len[(size_t)r];
        const uint8_t* p = got.data() + (size_t)r * SQ_X_PIECE + 8;
        all.insert(all.end(), p, p + std::min(n, head));
        if (n > head) { const uint8_t* q = rest_got.data() + (size_t)r * (size_t)rest; all.insert(all.end(), q, q + (n - head)); }
    }
    c->x_bytes += (int64_t)all.size();
    return sq_exchange_unpack(c, all.data(), len.data(), W);
}

int sq_exchange_stats(sq_ctx* c, int64_t* collectives, int64_t* bytes) {
    if (!c) return SQ_E_ARG;
    if (collectives) *collectives = c->x_collectives;
    if (bytes) *bytes = c->x_bytes;
    return SQ_OK;
}

}  // extern "C"
