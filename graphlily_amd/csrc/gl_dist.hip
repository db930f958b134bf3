// The one exchange step of the row-sharded (multi-GPU) path, in the C ABI: RCCL over xGMI.
//
// The reference is single-device.  SURVEY.md 8(e) defines the decomposition this build adds: contiguous row ranges,
// one per GPU / process; each iteration a rank produces its slice of the result and ONE all-gather rebuilds the full
// dense vector (BFS: bit vector; push iterations: sparse list) that is the next iteration's input.  graphlily_amd/dist.py
// does that with torch.distributed; these entry points do the same for C / C++ callers of the drop-in headers
// (SpMVModule::set_row_shard), on the library's stream, with no Python in the process.
//
// RCCL is loaded at run time (dlopen; an already loaded copy -- e.g. the one PyTorch ships -- is reused), so
// single-GPU users of the library need nothing beyond the HIP runtime.  Slices may have different lengths (row ranges
// are nnz-balanced): every rank sends its slice to every other rank and receives theirs inside one group call --
// point-to-point pushes over the fully connected xGMI links, no ring.
#include "gl_common.h"

#include <cstring>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <vector>

struct gl_dist_s {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int recorded = 0;                 // live gl_graphs that replay this communicator's exchanges (see gl_dist_destroy)
    uint32_t *d_counts = nullptr;     // world sparse-list heads (8 bytes each) for gl_dist_all_gather_sparse
    uint32_t *h_counts = nullptr;     // page-locked mirror
};

namespace gl {
namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names)   // a copy that is already in the process (PyTorch's) first: two RCCLs must not coexist
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!r.handle)
        for (const char *n : names)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.handle) {
        if ((r.handle = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL)) == nullptr) return r;
    }
#define GL_SYM(field, name)                                                     \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));       \
    if (!r.field) return r
    GL_SYM(GetUniqueId, "ncclGetUniqueId");
    GL_SYM(CommInitRank, "ncclCommInitRank");
    GL_SYM(CommDestroy, "ncclCommDestroy");
    GL_SYM(AllGather, "ncclAllGather");
    GL_SYM(Send, "ncclSend");
    GL_SYM(Recv, "ncclRecv");
    GL_SYM(GroupStart, "ncclGroupStart");
    GL_SYM(GroupEnd, "ncclGroupEnd");
    GL_SYM(GetErrorString, "ncclGetErrorString");
#undef GL_SYM
    r.ok = true;
    return r;
}

#define GL_NCCL(expr)                                                                                     \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess)                                                                            \
            return gl::set_error(GL_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, gl::rccl().GetErrorString(r_)); \
    } while (0)

int need_rccl(const char *who) {
    if (rccl().ok) return GL_OK;
    return set_error(GL_ERR_UNSUPPORTED, "%s: librccl.so could not be loaded (%s)", who, dlerror() ? dlerror() : "missing symbol");
}

// every rank's slice [lo[r], hi[r]) (BYTE offsets into `buf`, gl_dist_slice_plan) to every other rank, in place
// (buf2: a second vector of equal slices -- `each2` bytes per rank, rank r's at r * each2 -- in the SAME group: one operation)
int exchange_slices(gl_dist d, char *buf, const uint64_t *lo, const uint64_t *hi, char *buf2 = nullptr, uint64_t each2 = 0) {
    Rccl &R = rccl();
    hipStream_t s = ctx().stream;
    if (ctx().capturing) {
        std::vector<int *> &refs = ctx().capture_refs;
        bool seen = false;
        for (int *r : refs) seen |= r == &d->recorded;
        if (!seen) refs.push_back(&d->recorded);
    }
    if (d->world == 1) {
        // One GPU holds a world of one: nothing to exchange.  GRAPHLILY_DEBUG=dist_self_probe=1 still puts one RCCL
        // operation on the stream here (8 bytes of the slice sent to this same rank, into the communicator's scratch), so
        // that a one-GPU box exercises what a world of N relies on: RCCL calls recorded by a stream capture and replayed.
        static const bool probe = debug_knob("dist_self_probe", 0) != 0;
        if (probe && hi[0] - lo[0] >= 8) {
            GL_NCCL(R.GroupStart());
            GL_NCCL(R.Send(buf + lo[0], 8, ncclUint8, 0, d->comm, s));
            GL_NCCL(R.Recv(d->d_counts, 8, ncclUint8, 0, d->comm, s));
            GL_NCCL(R.GroupEnd());
        }
        return GL_OK;
    }
    const uint64_t mine = hi[d->rank] - lo[d->rank];
    GL_NCCL(R.GroupStart());
    for (int p = 0; p < d->world; p++) {
        if (p == d->rank) continue;
        const uint64_t theirs = hi[p] - lo[p];
        if (mine) GL_NCCL(R.Send(buf + lo[d->rank], mine, ncclUint8, p, d->comm, s));
        if (theirs) GL_NCCL(R.Recv(buf + lo[p], theirs, ncclUint8, p, d->comm, s));
        if (buf2 && each2) {
            GL_NCCL(R.Send(buf2 + (uint64_t)d->rank * each2, each2, ncclUint8, p, d->comm, s));
            GL_NCCL(R.Recv(buf2 + (uint64_t)p * each2, each2, ncclUint8, p, d->comm, s));
        }
    }
    GL_NCCL(R.GroupEnd());
    return GL_OK;
}

__global__ void dist_write_head_kernel(gl_idx_val *full, uint32_t total, float head_val) {
    full[0].index = total;
    full[0].val = head_val;
}

}  // namespace
}  // namespace gl

extern "C" {

int gl_dist_unique_id(void *id128) {
    GL_ARG(id128 != nullptr);
    int rc = gl::need_rccl("gl_dist_unique_id");
    if (rc != GL_OK) return rc;
    ncclUniqueId id;
    GL_NCCL(gl::rccl().GetUniqueId(&id));
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return GL_OK;
}

int gl_dist_init(gl_dist *comm, int rank, int world_size, const void *id128) {
    GL_REQUIRE_INIT();
    GL_ARG(comm != nullptr && id128 != nullptr && world_size >= 1 && rank >= 0 && rank < world_size);
    int rc = gl::need_rccl("gl_dist_init");
    if (rc != GL_OK) return rc;
    *comm = nullptr;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    gl_dist d = new gl_dist_s();
    d->rank = rank;
    d->world = world_size;
    ncclResult_t r = gl::rccl().CommInitRank(&d->comm, world_size, id, rank);   // on the device gl_init selected
    if (r != ncclSuccess) {
        delete d;
        return gl::set_error(GL_ERR_HIP, "gl_dist_init: ncclCommInitRank -> %s", gl::rccl().GetErrorString(r));
    }
    if (hipMalloc((void **)&d->d_counts, (size_t)world_size * 8u) != hipSuccess ||
        hipHostMalloc((void **)&d->h_counts, (size_t)world_size * 8u, hipHostMallocDefault) != hipSuccess) {
        gl_dist_destroy(d);
        return gl::set_error(GL_ERR_HIP, "gl_dist_init: out of memory");
    }
    *comm = d;
    return GL_OK;
}

int gl_dist_destroy(gl_dist d) {
    if (!d) return GL_OK;
    if (d->recorded > 0)
        return gl::set_error(GL_ERR_INVALID_ARG, "gl_dist_destroy: %d recorded graph(s) still replay this communicator's exchanges: "
                             "gl_graph_destroy them first (RCCL's destroy waits for them)", d->recorded);
    if (gl::ctx().initialized) (void)hipStreamSynchronize(gl::ctx().stream);
    if (d->comm && gl::rccl().ok) (void)gl::rccl().CommDestroy(d->comm);
    (void)hipFree(d->d_counts);
    (void)hipHostFree(d->h_counts);
    delete d;
    return GL_OK;
}

int gl_dist_rank(gl_dist d, int *rank, int *world_size) {
    GL_ARG(d != nullptr);
    if (rank) *rank = d->rank;
    if (world_size) *world_size = d->world;
    return GL_OK;
}

// Which bytes of the exchanged buffer rank r owns (sends to everybody, is received from by everybody):
//   GL_DIST_F32     in = world + 1 element bounds of a float vector           -> [4 lo, 4 hi)
//   GL_DIST_BITS    in = world + 1 ROW bounds of a bit vector, multiples of 32 (the last one may be anything: it is the
//                   end of the vector) -> whole 32-bit words [lo / 32, ceil(hi / 32)) x 4; a bound inside a word would
//                   make two ranks write that word
//   GL_DIST_SPARSE  in = world entry counts of the ranks' sparse lists -> rank r's entries follow the head element and
//                   the entries of the ranks before it: [8 (1 + sum_{q<r} count_q), 8 (1 + sum_{q<=r} count_q))
// Host arithmetic only (no device, no RCCL): what the three all-gathers below hand to the grouped send / receive.
int gl_dist_slice_plan(int kind, int world_size, const uint32_t *in, uint64_t *lo_bytes, uint64_t *hi_bytes) {
    GL_ARG(world_size >= 1 && in != nullptr && lo_bytes != nullptr && hi_bytes != nullptr);
    uint64_t total = 0;
    for (int r = 0; r < world_size; r++) {
        switch (kind) {
            case GL_DIST_F32:
                GL_ARG(in[r] <= in[r + 1]);
                lo_bytes[r] = 4ull * in[r];
                hi_bytes[r] = 4ull * in[r + 1];
                break;
            case GL_DIST_BITS:
                GL_ARG(in[r] <= in[r + 1]);
                GL_ARG(in[r] % 32u == 0 && (in[r + 1] % 32u == 0 || r + 1 == world_size));
                lo_bytes[r] = 4ull * (in[r] / 32u);
                hi_bytes[r] = 4ull * (((uint64_t)in[r + 1] + 31u) / 32u);
                break;
            case GL_DIST_SPARSE:
                lo_bytes[r] = 8ull * (1u + total);
                total += in[r];
                hi_bytes[r] = 8ull * (1u + total);
                break;
            default: return gl::set_error(GL_ERR_INVALID_ARG, "gl_dist_slice_plan: unknown kind %d", kind);
        }
    }
    return GL_OK;
}

int gl_dist_all_gather_f32(gl_dist d, float *d_full, const uint32_t *bounds) {
    GL_REQUIRE_INIT();
    GL_ARG(d != nullptr && d_full != nullptr && bounds != nullptr);
    std::vector<uint64_t> lo(d->world), hi(d->world);
    const int rc = gl_dist_slice_plan(GL_DIST_F32, d->world, bounds, lo.data(), hi.data());
    if (rc != GL_OK) return rc;
    return gl::exchange_slices(d, reinterpret_cast<char *>(d_full), lo.data(), hi.data());
}

int gl_dist_all_gather_bits_tally(gl_dist d, uint32_t *d_bits, const uint32_t *row_bounds, uint32_t *d_tally_slot, uint32_t bytes_per_rank) {
    GL_REQUIRE_INIT();
    GL_ARG(d != nullptr && d_bits != nullptr && row_bounds != nullptr && ((d_tally_slot != nullptr) == (bytes_per_rank > 0)));
    std::vector<uint64_t> lo(d->world), hi(d->world);
    const int rc = gl_dist_slice_plan(GL_DIST_BITS, d->world, row_bounds, lo.data(), hi.data());
    if (rc != GL_OK) return rc;
    if (!d_tally_slot) return gl::exchange_slices(d, reinterpret_cast<char *>(d_bits), lo.data(), hi.data());   // the bits alone
    return gl::exchange_slices(d, reinterpret_cast<char *>(d_bits), lo.data(), hi.data(), reinterpret_cast<char *>(d_tally_slot), bytes_per_rank);
}

int gl_dist_all_gather_sparse(gl_dist d, const gl_idx_val *d_local, gl_idx_val *d_full, uint32_t capacity, float head_val,
                              uint32_t *total_out) {
    GL_REQUIRE_INIT();
    GL_ARG(d != nullptr && d_local != nullptr && d_full != nullptr && (const void *)d_local != (const void *)d_full);
    int rc = gl::need_rccl("gl_dist_all_gather_sparse");
    if (rc != GL_OK) return rc;
    gl::Rccl &R = gl::rccl();
    hipStream_t s = gl::ctx().stream;
    // 1. every rank's head {count, -}: 8 bytes each, one all-gather; the host needs the counts to size the exchange
    if (d->world > 1) GL_NCCL(R.AllGather(d_local, d->d_counts, 8, ncclUint8, d->comm, s));
    else GL_HIP(hipMemcpyAsync(d->d_counts, d_local, 8, hipMemcpyDeviceToDevice, s));
    GL_HIP(hipMemcpyAsync(d->h_counts, d->d_counts, (size_t)d->world * 8u, hipMemcpyDeviceToHost, s));
    GL_HIP(hipStreamSynchronize(s));
    std::vector<uint64_t> lo(d->world), hi(d->world);
    std::vector<uint32_t> counts(d->world);
    uint64_t total = 0;
    for (int r = 0; r < d->world; r++) {
        counts[r] = d->h_counts[2 * r];
        total += counts[r];
    }
    rc = gl_dist_slice_plan(GL_DIST_SPARSE, d->world, counts.data(), lo.data(), hi.data());   // entries follow the head element
    if (rc != GL_OK) return rc;
    if (total > capacity) return gl::set_error(GL_ERR_INVALID_ARG, "gl_dist_all_gather_sparse: %llu entries exceed the capacity %u", (unsigned long long)total, capacity);
    // 2. my entries into my place of the full list, then the slices travel (rank order = ascending rows: the ranges are disjoint)
    const uint64_t mine = hi[d->rank] - lo[d->rank];
    if (mine) GL_HIP(hipMemcpyAsync(reinterpret_cast<char *>(d_full) + lo[d->rank], d_local + 1, mine, hipMemcpyDeviceToDevice, s));
    rc = gl::exchange_slices(d, reinterpret_cast<char *>(d_full), lo.data(), hi.data());
    if (rc != GL_OK) return rc;
    gl::dist_write_head_kernel<<<1, 1, 0, s>>>(d_full, (uint32_t)total, head_val);
    GL_LAUNCH_CHECK();
    if (total_out) *total_out = (uint32_t)total;
    return GL_OK;
}

}  // extern "C"
