// SpMSpV over a semiring for MI355X (gfx950): sparse y = mask( A_csc (x) sparse x ).
//
// Replaces SpMSpVModule::load_and_format_matrix -> formatCSC
// (module/spmspv_module.h:263-286, io/data_formatter.h:543-721) and kernel_spmspv
// (hw/kernel_spmspv_impl.h:448-562).
//
// Layout: the CSC is kept as indptr[num_cols+1] plus one packed 8-byte stream
// { row, val } in column order (the FPGA's 8-wide column packets padded with
// (0, zero) entries, data_formatter.h:674-680, are not needed: a wavefront reads
// any contiguous run of a column coalesced).
//
// Kernels per run (gl_spmspv_bin.h): BIN -- the products of the active columns, sorted by row tile in LDS, leave in runs into
// per-tile bins -- and FOLD -- one workgroup per row tile accumulates its bin in LDS and writes its piece of the ordered
// result list, with the mask, the fused sparse assign, the next-frontier bits and the driver's loop decision.  A vector of
// at most 1024 entries / 2048 products is ONE launch of one workgroup (spmspv_tiny_kernel); one whose columns hold a large
// part of the matrix is applied row-wise with an attached SpMV plan and folded from the dense accumulator.
#include "gl_common.h"
#include "gl_compact.h"
#include "gl_spmspv_bin.h"
#include "gl_spmv_plan.h"
#include "gl_bfs_shard.h"

#include <algorithm>
#include <cstring>
#include <vector>

struct gl_spmspv_plan_s {
    uint32_t num_rows = 0, num_cols = 0, row_begin = 0, row_end = 0;
    uint64_t nnz = 0;
    uint32_t *d_indptr = nullptr;  // num_cols + 1, offsets into d_stream
    uint2 *d_stream = nullptr;     // {row, val bits}
    float *d_acc = nullptr;        // dense accumulator over the shard's rows
    float acc_fill = 0.0f;
    bool acc_valid = false;        // d_acc is known to be all == acc_fill
    // propagation blocking (gl_spmspv_bin.h): row tiles, one bin per tile with room for every non-zero of the tile
    gl::TileMap tiles;
    bool binned = false;               // false: more tiles than the bin kernel has counters for -- products go to d_acc
    bool fold_tickets = false;         // more tiles than compute units: the fold hands its tiles out in arrival order
    uint2 *d_bins = nullptr;
    uint32_t *d_bin_base = nullptr;    // tiles + 1
    uint32_t *d_cursor = nullptr;      // tiles, zero between runs
    uint32_t *d_state = nullptr;       // tiles, zero between runs
    uint32_t *d_sync = nullptr;        // gl::kSyncWords, zero between runs
    unsigned long long *d_slices = nullptr;  // 2 x gl::kBinMaxSlices tagged slice sums of the bin kernel's rendezvous (by round parity)
    // a blocking caller's completion record: the fold's last workgroup stores seq << 32 | count (gl_spmspv_wait)
    unsigned long long *h_rec = nullptr;     // page-locked, device-visible
    uint32_t seq = 0;                        // of the last run that was given the record
    bool rec_pending = false;                // that run is the last one enqueued on this plan
    uint64_t rec_epoch = 0;                  // gl::graph_launches() when it was enqueued: a replayed graph may have rerun the plan since
    uint32_t nnz_hint = ~0u;                 // entries of the next run's vector, if a hint said so (sizes the bin grid)
    // direction switch inside the operator ((||,&&) only): a frontier whose columns hold more than 1/32 of the
    // matrix is cheaper to apply row-wise with the attached boolean SpMV plan than to scatter
    gl_spmv_plan pull = nullptr;        // not owned; boolean layout, serves (||,&&)
    gl_spmv_plan pull_arith = nullptr;  // not owned; general / pattern layout of the same matrix, serves (+,x)
    float *d_xdense = nullptr;          // the frontier as a dense vector for pull_arith (num_cols floats)
    uint32_t max_col_len = 0;           // longest column of the shard
    uint64_t frontier_hint = ~0ull;     // caller's upper bound on the next run's vector nnz (~0 = unknown)
    uint32_t *d_mode = nullptr;         // [0] 1 = this run goes row-wise, [1] block ticket, [2..3] work counter
    bool last_decided_on_device = false;   // the last run launched the decision kernel (d_mode[0] is its verdict)
    // gl_bfs_bits_push_step: the chunks of every long column as a static list {column, first entry, count, -}; a push
    // step tests the frontier bit of each chunk's column instead of queueing chunks at run time (no second launch)
    uint4 *d_long_chunks = nullptr;
    uint32_t n_long_chunks = 0;
    // gl_spmspv_plan_hint_tiny (one-shot): the caller expects the next run's vector to be tiny -- then the run is ONE launch
    bool tiny_hint = false;
    // gl_spmspv_plan_hint_work (one-shot): the caller knows the next run's vector -- the non-zeros of its columns and the longest
    // of them; ~0 = unknown
    uint64_t work_hint = ~0ull;
    uint32_t longest_hint = 0;
    const void *bfs_rows_plan = nullptr;   // the SpMV plan whose rows the last gl_bfs_bits_push_step could scan bottom-up (or null)
    uint32_t *d_bfs_acc = nullptr;   // kBfsAccSlots x 32 words: the push step's totals, spread over 64 lines (see bfs_push_bits_kernel)
    uint64_t device_bytes = 0;
};

namespace gl {

// Live SpMSpV plans, so that destroying an SpMV plan detaches it wherever it is attached (gl_spmspv_plan_attach_pull
// does not own the plan; a module that re-formats or dies must not leave a dangling pointer behind).
static std::vector<gl_spmspv_plan> &live_spmspv_plans() {
    static std::vector<gl_spmspv_plan> v;
    return v;
}

void spmspv_detach_everywhere(gl_spmv_plan dying) {
    for (gl_spmspv_plan q : live_spmspv_plans()) {
        if (q->pull == dying) q->pull = nullptr;
        if (q->pull_arith == dying) q->pull_arith = nullptr;
    }
}

constexpr uint32_t kBfsAccSlots = 64;  // accumulator lines of the bit-frontier BFS push step (a power of two)

// exclusive prefix of v over the 256 threads of the block; total broadcast through *total
__device__ __forceinline__ uint32_t block_exclusive_256(uint32_t v, uint32_t *s_wave, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (uint32_t dlt = 1; dlt < 64; dlt <<= 1) {
        uint32_t up = __shfl_up(incl, dlt);
        if (lane >= dlt) incl += up;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t c = s_wave[k];
        if (k < w) before += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return before + incl - v;
}

// work of this run = sum of the lengths of the frontier's columns; the last block to finish sets the mode
__global__ __launch_bounds__(256) void spmspv_work_kernel(const gl_idx_val *__restrict__ vec, const uint32_t *__restrict__ indptr,
                                                          uint32_t num_cols, uint32_t *__restrict__ mode, uint64_t threshold) {
    __shared__ unsigned long long s_sum[4];
    const uint32_t vnnz = vec[0].index;
    unsigned long long w = 0;
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < vnnz; e += gridDim.x * 256u) {
        const gl_idx_val iv = vec[1u + e];
        if (iv.index < num_cols && iv.val != 0.0f) w += indptr[iv.index + 1u] - indptr[iv.index];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) w += __shfl_down(w, d);
    if ((threadIdx.x & 63u) == 0) s_sum[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *total = reinterpret_cast<unsigned long long *>(mode + 2);
        const unsigned long long mine = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        if (mine) atomicAdd(total, mine);
        __threadfence();
        if (atomicAdd(&mode[1], 1u) == gridDim.x - 1u) {   // last block: every partial sum has landed
            const unsigned long long t = atomicAdd(total, 0ull);
            mode[0] = t > threshold ? 1u : 0u;
            mode[1] = 0u;
            *total = 0ull;
        }
    }
}

// row-wise path: the frontier as a bit vector (the caller has zeroed `bits`)
__global__ __launch_bounds__(256) void spmspv_frontier_bits_kernel(const gl_idx_val *__restrict__ vec, uint32_t num_cols,
                                                                   uint32_t *__restrict__ bits, const uint32_t *__restrict__ mode) {
    if (!mode[0]) return;
    const uint32_t vnnz = vec[0].index;
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < vnnz; e += gridDim.x * 256u) {
        const gl_idx_val iv = vec[1u + e];
        if (iv.index < num_cols && iv.val != 0.0f) atomicOr(&bits[iv.index >> 5], 1u << (iv.index & 31u));
    }
}

// row-wise (+,x) path: zero the dense vector (only when the run goes row-wise) ...
__global__ __launch_bounds__(256) void spmspv_clear_dense_kernel(float4 *__restrict__ dense4, uint32_t n4, float fill,
                                                                 const uint32_t *__restrict__ mode) {
    if (!mode[0]) return;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) dense4[i] = make_float4(fill, fill, fill, fill);
}

// ... and scatter the frontier into it
__global__ __launch_bounds__(256) void spmspv_frontier_dense_kernel(const gl_idx_val *__restrict__ vec, uint32_t num_cols,
                                                                    float *__restrict__ dense, const uint32_t *__restrict__ mode) {
    if (!mode[0]) return;
    const uint32_t vnnz = vec[0].index;
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < vnnz; e += gridDim.x * 256u) {
        const gl_idx_val iv = vec[1u + e];
        if (iv.index < num_cols) dense[iv.index] = iv.val;
    }
}

// ------------------------------------------------------------------ BFS push step on a bit frontier (gl_bfs_bits_push_step):
// BfsPushArgs, bfs_claim and bfs_candidate are in gl_bfs_shard.h (the one-launch shard step runs the same bodies)
__global__ __launch_bounds__(256) void bfs_push_bits_kernel(BfsPushArgs a) {
    __shared__ uint32_t s_words[256];
    __shared__ uint32_t s_start[256];
    __shared__ uint32_t s_deg[256];
    __shared__ uint32_t s_task[257];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_fresh, s_nhit;
    __shared__ unsigned long long s_work, s_work_rows;
    if (a.bits_spare)
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < a.words; i += gridDim.x * 256u) a.bits_spare[i] = 0u;
    if (a.c.finished()) return;
    const bool scatter = a.c.scatters();
    if (!scatter && !(a.row_idx && a.c.bottom_up())) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.c.record_mode(scatter ? 1u : 3u);
    if (threadIdx.x == 0) {
        s_fresh = 0u;
        s_nhit = 0u;
        s_work = 0ull;
        s_work_rows = 0ull;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t fresh = 0u, work = 0u, work_rows = 0u;
    if (!scatter) {
        // ---- bottom-up pull (the slot pulls, or pushes a heavy frontier, and few non-zeros are left in unreached rows):
        // a thread per row not reached yet looks through the row, four entries per step, until it finds a neighbour in the
        // frontier; 64 rows per wavefront = one word of the next frontier.  Same result as the streaming pull step
        // (masked (||,&&) SpMV + assign, app/bfs.h:118-123), at the cost of the unreached rows instead of the matrix.
        // (shard bounds are multiples of 64 rows: every 64-bit word of the next frontier has one writer)
        const uint32_t nwords64 = (a.row_end + 63u) >> 6;
        for (uint32_t wd = (a.row_begin >> 6) + blockIdx.x * 4u + wave; wd < nwords64; wd += gridDim.x * 4u) {
            const uint32_t row = wd * 64u + lane;
            const bool live = row < a.row_end && a.dist[row] == 0.0f;
            uint32_t beg = 0, end = 0;
            if (live) {
                beg = a.row_ptr[row - a.row_begin] - a.nz_base;
                end = a.row_ptr[row - a.row_begin + 1u] - a.nz_base;
            }
            const uint32_t len = end - beg;
            bool hit = false;
            for (int step = 0; step < 8 && __any(!hit && beg < end); step++) {
                if (!hit && beg < end) {
                    uint32_t c[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) c[u] = beg + u < end ? a.row_idx[beg + u] : 0xffffffffu;
                    uint32_t any = 0u;
#pragma unroll
                    for (int u = 0; u < 4; u++) any |= c[u] < a.num_cols ? (a.bits_in[c[u] >> 5] >> (c[u] & 31u)) & 1u : 0u;
                    hit = any != 0u;
                    beg += 4u;
                }
            }
            // rows still undecided after 32 entries are finished by the whole wavefront, 256 entries per step: a hub row
            // the BFS never reaches (another component) must not keep one thread busy for its 100 K entries
            for (uint64_t pending = __ballot(!hit && beg < end); pending; pending &= pending - 1ull) {
                const int src = __ffsll((unsigned long long)pending) - 1;
                const uint32_t b = __shfl(beg, src), e = __shfl(end, src);
                bool found = false;
                for (uint32_t base = b; base < e && !found; base += 256u) {
                    uint32_t any = 0u;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t k = base + 64u * u + lane;
                        const uint32_t c = k < e ? a.row_idx[k] : 0xffffffffu;
                        any |= c < a.num_cols ? (a.bits_in[c >> 5] >> (c & 31u)) & 1u : 0u;
                    }
                    found = __any(any != 0u);
                }
                if ((int)lane == src) hit = found;
            }
            if (hit) {
                a.dist[row] = a.level;
                if (!a.deferred) {
                    fresh += 1u;
                    work_rows += len;
                    if (row < a.num_cols) work += a.indptr[row + 1u] - a.indptr[row];
                }
            }
            const uint64_t m = __ballot(hit);
            if (lane == 0) reinterpret_cast<uint64_t *>(a.bits_out)[wd] = m;
        }
    } else {

    // this workgroup's slice of the frontier words: read 256 words at a time (one load per thread), then 8 words = 256
    // columns per batch out of LDS; empty pieces cost one barrier
    const uint32_t per = ((a.col_words + gridDim.x - 1u) / gridDim.x + 7u) & ~7u;
    const uint32_t w_begin = blockIdx.x * per, w_end = min(a.col_words, w_begin + per);
    for (uint32_t wp = w_begin; wp < w_end; wp += 256u) {
        const uint32_t mine = (wp + threadIdx.x < w_end) ? a.bits_in[wp + threadIdx.x] : 0u;
        __syncthreads();                      // the previous piece's batches are done with s_words
        s_words[threadIdx.x] = mine;
        if (!__syncthreads_or(mine != 0u)) continue;
        const uint32_t nw = min(256u, w_end - wp);
        for (uint32_t b0 = 0; b0 < nw; b0 += 8u) {
            const uint32_t word = s_words[b0 + (threadIdx.x >> 5)];
            {   // block-uniform: any of the batch's 8 words set?
                uint32_t any = 0u;
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) any |= s_words[b0 + k];
                if (!any) continue;
            }
            const uint32_t col = (wp + b0 + (threadIdx.x >> 5)) * 32u + (threadIdx.x & 31u);
            uint32_t start = 0, deg = 0;
            if (((word >> (threadIdx.x & 31u)) & 1u) && col < a.num_cols) {
                start = a.indptr[col];
                deg = a.indptr[col + 1u] - start;
                if (deg >= kBfsChunk) deg = 0u;       // served from the chunk list below
            }
            uint32_t ttotal;
            const uint32_t toff = block_exclusive_256((deg + 63u) >> 6, s_wave, &ttotal);
            s_start[threadIdx.x] = start;
            s_deg[threadIdx.x] = deg;
            s_task[threadIdx.x] = toff;
            if (threadIdx.x == 255) s_task[256] = ttotal;
            __syncthreads();
            // two wave tasks of <= 64 entries per step: the stream loads, the distance loads and the atomics of the two are
            // in flight together (a task is three dependent memory round trips)
            for (uint32_t t = wave; t < ttotal; t += 8u) {
                uint2 rv[2];
                bool valid[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t tu = min(t + 4u * u, ttotal - 1u);
                    uint32_t lo = 0, hi = 255;   // largest j with s_task[j] <= tu (wave-uniform)
#pragma unroll
                    for (int it = 0; it < 8; it++) {
                        const uint32_t mid = (lo + hi + 1u) >> 1;
                        if (s_task[mid] <= tu) lo = mid; else hi = mid - 1u;
                    }
                    const uint32_t j = __builtin_amdgcn_readfirstlane(lo);
                    const uint32_t item = ((tu - s_task[j]) << 6) + lane;
                    valid[u] = (t + 4u * u < ttotal) && item < s_deg[j];
                    rv[u] = valid[u] ? load_stream_nt(a.stream + s_start[j] + item) : make_uint2(0u, 0u);
                }
                const bool c0 = bfs_candidate(a, valid[0], rv[0]), c1 = bfs_candidate(a, valid[1], rv[1]);
                bfs_claim(a, c0, rv[0].x, fresh, work, work_rows);
                bfs_claim(a, c1, rv[1].x, fresh, work, work_rows);
            }
            __syncthreads();
        }
    }
    // chunks of long columns: every thread tests the frontier bit of one chunk, the hits are processed by the whole workgroup
    __syncthreads();
    // (thread t of workgroup b tests chunk t * #workgroups + b of the pass: the chunks of one hub column, neighbours in the
    // list, go to different workgroups -- dealt 256 in a row to one workgroup, a hub's 25 chunks ran one after the other:
    // 784 us for a 330-vertex frontier of the pokec stand-in)
    for (uint32_t q0 = 0; q0 < a.nchunks; q0 += gridDim.x * 256u) {
        const uint32_t q = q0 + threadIdx.x * gridDim.x + blockIdx.x;
        if (q < a.nchunks) {
            const uint32_t col = a.chunks[q].x;
            if ((a.bits_in[col >> 5] >> (col & 31u)) & 1u) s_start[atomicAdd(&s_nhit, 1u)] = q;
        }
        __syncthreads();
        const uint32_t nhit = s_nhit;
        __syncthreads();
        if (threadIdx.x == 0) s_nhit = 0u;
        for (uint32_t h = 0; h < nhit; h++) {
            const uint4 ch = a.chunks[s_start[h]];
            for (uint32_t k = threadIdx.x; k < ch.z; k += 512u) {
                const bool v1 = k + 256u < ch.z;
                const uint2 r0 = load_stream_nt(a.stream + ch.y + k);
                const uint2 r1 = v1 ? load_stream_nt(a.stream + ch.y + k + 256u) : make_uint2(0u, 0u);
                const bool c0 = bfs_candidate(a, true, r0), c1 = bfs_candidate(a, v1, r1);
                bfs_claim(a, c0, r0.x, fresh, work, work_rows);
                bfs_claim(a, c1, r1.x, fresh, work, work_rows);
            }
        }
        __syncthreads();
    }
    }   // scatter
    if (a.deferred) return;
    // totals of the step
    unsigned long long work64 = work, rows64 = work_rows;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        fresh += __shfl_down(fresh, d);
        work64 += __shfl_down(work64, d);
        rows64 += __shfl_down(rows64, d);
    }
    __syncthreads();
    if (lane == 0 && fresh) {
        atomicAdd(&s_fresh, fresh);
        atomicAdd(&s_work, work64);
        atomicAdd(&s_work_rows, rows64);
    }
    __syncthreads();
    // The step's totals go to one of 64 accumulator lines (thousands of workgroups ending on ONE word serialise: 45 us for
    // an empty step on 2048 workgroups); the pull step of the same slot -- launched right behind, gated off -- adds them
    // up and takes the step's decisions (bool_plan_bfs_step), so no workgroup here has to find out that it is the last.
    if (threadIdx.x == 0 && s_fresh) {
        uint32_t *line = a.acc + 32u * (blockIdx.x & (kBfsAccSlots - 1u));
        atomicAdd(line, s_fresh);
        atomicAdd(reinterpret_cast<unsigned long long *>(line + 2), s_work);
        if (s_work_rows) atomicAdd(reinterpret_cast<unsigned long long *>(line + 6), s_work_rows);
    }
}

// compaction source over the dense accumulator
template <int MASK, bool BITS = false>   // BITS: the integer value types compare bit patterns (zero may be 0xffffffff, a NaN as a float)
struct AccSource {
    float *acc;
    const float *mask;
    uint32_t nrows;      // rows in the shard
    uint32_t row_begin;
    float zero;
    float *assign;       // gl_spmspv_run_assign: assign[index] = assign_val for every emitted entry (or null)
    float assign_val;
    __device__ uint32_t size() const { return nrows; }
    __device__ bool get(uint32_t i, gl_idx_val &out) const {
        float v = acc[i];
        if (BITS ? (__float_as_uint(v) == __float_as_uint(zero)) : (v == zero)) return false;  // checkout_results: dense_data != zero (kernel_spmspv_impl.h:199-226)
        if (MASK != GL_NOMASK) {
            // write_back_gmem compares the mask with `zero` (kernel_spmspv_impl.h:262-283)
            const float m = mask[row_begin + i];
            const bool eq = BITS ? (__float_as_uint(m) == __float_as_uint(zero)) : (m == zero);
            if (MASK == GL_MASK_WRITETOZERO ? !eq : eq) return false;
        }
        out.index = row_begin + i;
        out.val = v;
        return true;
    }
    __device__ void consumed(uint32_t i) const {
        if (BITS ? (__float_as_uint(acc[i]) != __float_as_uint(zero)) : (acc[i] != zero)) acc[i] = zero;
    }
    // the entry's own row: no other thread reads or writes assign[item.index] in this pass
    __device__ void emitted(const gl_idx_val &item) const {
        if (assign) assign[item.index] = assign_val;
    }
};

// ------------------------------------------------------------------ tiny runs: one launch (gl_spmspv_plan_hint_tiny)
// A run whose vector holds <= kTinyVec entries with <= kTinyWork non-zeros in their columns (1024 / 2048: what one compute
// unit gets through in a few microseconds -- with 8192 a run took longer than the four launches) is done by ONE workgroup:
// scatter with atomics that return the old value -- the first product to reach a row (old == fill) appends the row to an LDS
// list --, sort the list, and emit it in ascending row order through the same AccSource as the compaction passes (mask, assign,
// next-frontier bits, accumulator reset).  The four dependent launches of the general path (scatter, queue, count, write) cost
// ~50 us per blocking call whatever the vector holds; this one ~20.  The kernel checks the two bounds itself on the vector it
// actually finds: a vector that is not tiny after all is still computed correctly, by the same workgroup, slowly (a dense
// pass over the shard's rows) -- the hint is about speed, never about results.
constexpr uint32_t kTinyVec = 1024, kTinyWork = 2048, kTinyThreads = 1024;

struct TinyArgs {
    const uint32_t *indptr;
    const uint2 *stream;
    const gl_idx_val *vec;
    float *acc;
    uint32_t row_begin, nrows, num_cols;
    gl_idx_val *out;
    float head_val;
    uint32_t zero_bits;
    unsigned long long *host_rec;   // a blocking caller's completion record (gl_spmspv_wait): seq << 32 | count, or null
    uint32_t seq;
    uint32_t bucket_shift;          // (nrows - 1) >> bucket_shift < 2048: the sort's buckets
};

// the one workgroup has written everything: tell a blocking caller
__device__ __forceinline__ void tiny_report(const TinyArgs &a, uint32_t count) {
    if (!a.host_rec) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(a.host_rec, ((unsigned long long)a.seq << 32) | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ uint32_t atomic_min_float_old(float *addr, float v) {
    if (!(__float_as_uint(v) >> 31)) return (uint32_t)atomicMin((int *)addr, __float_as_int(v));
    return atomicMax((unsigned int *)addr, __float_as_uint(v));
}

// scatter one product; true when this was the first product to reach the row (the atomic saw the fill value)
template <int OP>
__device__ __forceinline__ bool tiny_scatter(float *acc, uint32_t row, float a, float xv, uint32_t zero_bits) {
    if (OP == GL_OP_MULADD) return __float_as_uint(unsafeAtomicAdd(&acc[row], a * xv)) == zero_bits;
    if (OP == GL_OP_ANDOR) {
        if (a == 0.0f || xv == 0.0f) return false;
        return atomicExch(reinterpret_cast<unsigned int *>(&acc[row]), __float_as_uint(1.0f)) == zero_bits;
    }
    float incr;   // the saturating add of the (min,+) PE, as in scatter_one
    if (a > kFloatInf || xv > kFloatInf) {
        incr = kFloatInf;
    } else {
        incr = a + xv;
        if (incr > kFloatInf) incr = kFloatInf;
    }
    return atomic_min_float_old(&acc[row], incr) == zero_bits;
}

template <int OP, typename Src>
__global__ __launch_bounds__(kTinyThreads) void spmspv_tiny_kernel(TinyArgs a, Src src) {
    __shared__ uint32_t s_start[kTinyVec];
    __shared__ uint32_t s_pref[kTinyVec + 1];
    __shared__ float s_val[kTinyVec];
    __shared__ __attribute__((aligned(16))) uint32_t s_key[kTinyWork + 4];
    __shared__ uint32_t s_tmp[2048];
    __shared__ uint32_t s_hist[2048], s_base[2048];
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_cnt;
    const uint32_t tid = threadIdx.x;
    const uint32_t vnnz = a.vec[0].index;
    if (tid == 0) s_cnt = 0u;

    // ---- the vector's columns and the exclusive prefix of their lengths (first kTinyVec entries)
    uint32_t start = 0, deg = 0;
    float xv = 0.0f;
    if (tid < vnnz) {
        const gl_idx_val iv = a.vec[1u + tid];
        if (iv.index < a.num_cols) {
            start = a.indptr[iv.index];
            deg = a.indptr[iv.index + 1u] - start;
            xv = iv.val;
        }
    }
    uint32_t work;
    const uint32_t pre = block_exclusive_1024(deg, s_wave, &work);
    s_start[tid] = start;
    s_pref[tid] = pre;
    s_val[tid] = xv;
    if (tid == kTinyThreads - 1u) s_pref[kTinyVec] = work;
    __syncthreads();

    if (vnnz > kTinyVec || work > kTinyWork) {
        // ---- not tiny after all (a stale hint): everything by this one workgroup, slowly but correctly.  Scatter batch
        // by batch, then an ordered dense pass over the shard's rows.
        for (uint32_t b0 = 0; b0 < vnnz; b0 += kTinyThreads) {
            __syncthreads();
            uint32_t st = 0, dg = 0;
            float x = 0.0f;
            if (b0 + tid < vnnz) {
                const gl_idx_val iv = a.vec[1u + b0 + tid];
                if (iv.index < a.num_cols) {
                    st = a.indptr[iv.index];
                    dg = a.indptr[iv.index + 1u] - st;
                    x = iv.val;
                }
            }
            uint32_t wk;
            const uint32_t pr = block_exclusive_1024(dg, s_wave, &wk);
            s_start[tid] = st;
            s_pref[tid] = pr;
            s_val[tid] = x;
            if (tid == kTinyThreads - 1u) s_pref[kTinyVec] = wk;
            __syncthreads();
            for (uint32_t item = tid; item < wk; item += kTinyThreads) {
                uint32_t lo = 0, hi = kTinyVec - 1u;   // largest j with s_pref[j] <= item
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    if (s_pref[mid] <= item) lo = mid; else hi = mid - 1u;
                }
                const uint2 rv = load_stream_nt(a.stream + s_start[lo] + (item - s_pref[lo]));
                (void)tiny_scatter<OP>(a.acc, rv.x - a.row_begin, __uint_as_float(rv.y), s_val[lo], a.zero_bits);
            }
        }
        __threadfence();
        __syncthreads();
        uint32_t pos = 0;
        for (uint32_t base = 0; base < a.nrows; base += kTinyThreads) {
            const uint32_t i = base + tid;
            gl_idx_val item;
            bool keep = false;
            if (i < a.nrows) {
                keep = src.get(i, item);
                src.consumed(i);
            }
            uint32_t tot;
            const uint32_t rank = block_exclusive_1024(keep ? 1u : 0u, s_wave, &tot);
            if (keep) {
                a.out[1u + pos + rank] = item;
                src.emitted(item);
            }
            pos += tot;
        }
        if (tid == 0) {
            a.out[0].index = pos;
            a.out[0].val = a.head_val;
        }
        tiny_report(a, pos);
        return;
    }

    // ---- scatter; rows reached for the first time go to the candidate list.  work <= 2 * kTinyThreads: a thread has at most
    // two products, whose stream loads and atomics are issued together (each is two dependent round trips)
    {
        uint2 rv[2];
        float xs[2];
        bool has[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t item = tid + u * kTinyThreads;
            has[u] = item < work;
            uint32_t lo = 0, hi = kTinyVec - 1u;   // largest j with s_pref[j] <= item (zero-length columns share a prefix:
            while (lo < hi) {                      // the largest such j is the one that owns the item)
                const uint32_t mid = (lo + hi + 1u) >> 1;
                if (s_pref[mid] <= item) lo = mid; else hi = mid - 1u;
            }
            xs[u] = s_val[lo];
            rv[u] = has[u] ? load_stream_nt(a.stream + s_start[lo] + (item - s_pref[lo])) : make_uint2(0u, 0u);
        }
        bool first[2];
#pragma unroll
        for (int u = 0; u < 2; u++)
            first[u] = has[u] && tiny_scatter<OP>(a.acc, rv[u].x - a.row_begin, __uint_as_float(rv[u].y), xs[u], a.zero_bits);
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (first[u]) s_key[atomicAdd(&s_cnt, 1u)] = rv[u].x - a.row_begin;
    }
    __syncthreads();
    const uint32_t C = s_cnt;

    // ---- ascending rows.  One counting pass over 2048 buckets of 1 << bucket_shift consecutive rows -- the rows a tiny run reaches
    // are spread out, a bucket holds a key or two --, then every key ranks itself among its bucket's keys: five barriers and a
    // few dozen instructions per thread.  (Round 4's phase stamps: the O(C^2) rank sort this replaces was 13 us of the kernel's
    // 20 on 394 keys -- the GPU's clocks are low between blocking calls, instructions are what costs.)  Worst case, every key in
    // one bucket (millions of rows, all touched rows adjacent): the same C^2 compares as before.
    {
        const uint32_t sh = a.bucket_shift;
        s_hist[tid] = 0u;
        s_hist[tid + kTinyThreads] = 0u;
        __syncthreads();
        uint32_t kk[2] = {0u, 0u}, rr[2] = {0u, 0u};
#pragma unroll
        for (uint32_t u = 0; u < 2u; u++) {
            const uint32_t idx = tid + u * kTinyThreads;
            if (idx < C) {
                kk[u] = s_key[idx];
                rr[u] = atomicAdd(&s_hist[kk[u] >> sh], 1u);
            }
        }
        __syncthreads();
        const uint32_t c0 = s_hist[2u * tid], c1 = s_hist[2u * tid + 1u];
        uint32_t tot;
        const uint32_t before = block_exclusive_1024(c0 + c1, s_wave, &tot);
        s_base[2u * tid] = before;
        s_base[2u * tid + 1u] = before + c0;
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < 2u; u++)
            if (tid + u * kTinyThreads < C) s_tmp[s_base[kk[u] >> sh] + rr[u]] = kk[u];
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < 2u; u++) {
            const uint32_t idx = tid + u * kTinyThreads;
            if (idx < C) {
                const uint32_t k = s_tmp[idx], b = k >> sh, lo = s_base[b], n = s_hist[b];
                uint32_t r = 0;
                for (uint32_t q = lo; q < lo + n; q++) {
                    const uint32_t o = s_tmp[q];
                    r += (o < k || (o == k && q < idx)) ? 1u : 0u;
                }
                s_key[lo + r] = k;
            }
        }
        __syncthreads();
    }

    // ---- emission in row order: duplicates (a row that went back to the fill value and was reached again) are neighbours
    uint32_t pos = 0;
    for (uint32_t base = 0; base < C; base += kTinyThreads) {
        const uint32_t idx = base + tid;
        gl_idx_val item;
        bool keep = false;
        if (idx < C) {
            const uint32_t i = s_key[idx];
            if (idx == 0u || s_key[idx - 1u] != i) {
                keep = src.get(i, item);
                src.consumed(i);
            }
        }
        uint32_t tot;
        const uint32_t rank = block_exclusive_1024(keep ? 1u : 0u, s_wave, &tot);
        if (keep) {
            a.out[1u + pos + rank] = item;
            src.emitted(item);
        }
        pos += tot;
    }
    if (tid == 0) {
        a.out[0].index = pos;
        a.out[0].val = a.head_val;
    }
    tiny_report(a, pos);
}

template <int OP, int MASK>
static int launch_tiny(const TinyArgs &a, const float *mask, float zero, float *inout, float val, hipStream_t s) {
    AccSource<MASK> src{a.acc, mask, a.nrows, a.row_begin, zero, inout, val};
    spmspv_tiny_kernel<OP, AccSource<MASK>><<<1, kTinyThreads, 0, s>>>(a, src);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

template <int OP>
static int launch_tiny_mask(int mask_type, const TinyArgs &a, const float *mask, float zero, float *inout, float val, hipStream_t s) {
    switch (mask_type) {
        case GL_NOMASK: return launch_tiny<OP, GL_NOMASK>(a, mask, zero, inout, val, s);
        case GL_MASK_WRITETOZERO: return launch_tiny<OP, GL_MASK_WRITETOZERO>(a, mask, zero, inout, val, s);
        default: return launch_tiny<OP, GL_MASK_WRITETOONE>(a, mask, zero, inout, val, s);
    }
}

// see gl_spmspv_run's test hook: what a rendezvous poll that gives up stores (gl_spmspv_bin.h kSyncErr)
__global__ void spmspv_inject_timeout_kernel(uint32_t *sync) { sync[kSyncErr] = sync[kSyncGen]; }

template <int OPX>
static int launch_fold_op(const FoldArgs &f, hipStream_t s) {
    using T = typename Tile<OPX>::T;
    const size_t lds = (size_t)f.tiles.rows * sizeof(T);
    static bool attr_set = false;
    if (!attr_set) {
        GL_HIP(hipFuncSetAttribute((const void *)spmspv_fold_kernel<OPX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kFoldMaxRows * 8u)));
        attr_set = true;
    }
    spmspv_fold_kernel<OPX><<<f.tiles.count, kFoldThreads, lds, s>>>(f);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

static int launch_fold(int opx, const FoldArgs &f, hipStream_t s) {
    switch (opx) {
        case GL_OP_MULADD: return launch_fold_op<GL_OP_MULADD>(f, s);
        case GL_OP_ANDOR: return launch_fold_op<GL_OP_ANDOR>(f, s);
        case GL_OP_ADDMIN: return launch_fold_op<GL_OP_ADDMIN>(f, s);
        case kOpU32MulAdd: return launch_fold_op<kOpU32MulAdd>(f, s);
        case kOpU32AndOr: return launch_fold_op<kOpU32AndOr>(f, s);
        case kOpU32AddMin: return launch_fold_op<kOpU32AddMin>(f, s);
        case kOpFixMulAdd: return launch_fold_op<kOpFixMulAdd>(f, s);
        case kOpFixAndOr: return launch_fold_op<kOpFixAndOr>(f, s);
        case kOpFixAddMin: return launch_fold_op<kOpFixAddMin>(f, s);
        default: return set_error(GL_ERR_UNSUPPORTED, "gl_spmspv_run: unknown semiring / value type code %d", opx);
    }
}

// plan creation: non-zeros per row tile = the capacity of the tile's bin
__global__ __launch_bounds__(1024) void spmspv_tile_count_kernel(const uint2 *__restrict__ stream, uint64_t n, uint32_t row_begin, TileMap tiles,
                                                                 uint32_t *__restrict__ counts) {
    __shared__ uint32_t s_cnt[kBinMaxTiles];
    for (uint32_t i = threadIdx.x; i < kBinMaxTiles; i += 1024u) s_cnt[i] = 0u;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 1024u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 1024u)
        atomicAdd(&s_cnt[tiles.of(stream[i].x - row_begin)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < tiles.count; i += 1024u)
        if (s_cnt[i]) atomicAdd(&counts[i], s_cnt[i]);
}

// rows per tile: one tile per compute unit where the shard has rows enough (every tile is folded by one workgroup, all at
// the same time), a multiple of 64, at most kFoldMaxRows (the tile's 8-byte accumulators fill 128 KB of LDS)
static TileMap choose_tiles(uint32_t nrows, uint32_t num_cus) {
    TileMap tm;
    uint32_t R = cdiv(std::max<uint32_t>(nrows, 1u), std::max<uint32_t>(num_cus, 1u));
    R = std::min<uint32_t>(std::max<uint32_t>((R + 63u) & ~63u, 64u), kFoldMaxRows);
    const long forced = debug_knob("spmspv_tile_rows", 0);   // tests: many small tiles on a small matrix
    if (forced >= 64) R = std::min<uint32_t>(((uint32_t)forced + 63u) & ~63u, kFoldMaxRows);
    for (;;) {
        tm.rows = R;
        tm.count = std::max<uint32_t>(cdiv(nrows, R), 1u);
        uint32_t sh = 0;
        while ((2u << sh) <= R) sh++;                 // floor(log2 R)
        tm.shift = sh;
        if ((R & (R - 1u)) == 0u) {
            tm.magic = 0u;
            return tm;
        }
        tm.magic = (uint32_t)((1ull << (32u + sh)) / R) + 1u;   // < 2^32: R > 2^sh
        bool exact = true;
        for (uint32_t t = 1; t <= tm.count && exact; t++) {
            const uint64_t b = (uint64_t)t * R;
            if (b - 1u <= 0xffffffffull) exact = tm.of((uint32_t)(b - 1u)) == t - 1u;
            if (exact && b <= 0xffffffffull && b < (uint64_t)nrows + R) exact = tm.of((uint32_t)b) == t;
        }
        if (exact) return tm;
        uint32_t p2 = 64u;                            // (never seen: fall back to a power of two)
        while (p2 < R) p2 <<= 1;
        R = std::min<uint32_t>(p2, kFoldMaxRows);
    }
}

}  // namespace gl

extern "C" {

int gl_spmspv_plan_create(gl_spmspv_plan *plan, uint32_t num_rows, uint32_t num_cols,
                          const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data,
                          uint32_t row_begin, uint32_t row_end) {
    GL_TRACE();
    GL_REQUIRE_INIT();
    GL_ARG(plan != nullptr && h_indptr != nullptr);
    GL_ARG(row_begin <= row_end && row_end <= num_rows);
    const uint64_t nnz_all = h_indptr[num_cols];
    GL_ARG(nnz_all == 0 || (h_indices != nullptr && h_data != nullptr));
    const bool whole = (row_begin == 0 && row_end == num_rows);

    for (uint32_t c = 0; c < num_cols; c++) GL_ARG(h_indptr[c + 1] >= h_indptr[c] && h_indptr[c + 1] <= nnz_all);
    std::vector<uint32_t> indptr(num_cols + 1ull);
    std::vector<uint2> stream;
    // large matrices: the stream is built on the device from the uploaded CSC (gl_format.hip); the host loop below
    // is the same thing, serially
    const bool on_device = nnz_all > 0 && gl::format_on_device(0u, nnz_all);
    uint32_t *dev_indptr = nullptr;
    uint2 *dev_stream = nullptr;
    if (on_device) {
        const int frc = gl::fmt_spmspv_stream(num_rows, num_cols, h_indptr, h_indices, h_data, row_begin, row_end, &dev_indptr, &dev_stream, indptr);
        if (frc != GL_OK) return frc;
    } else {
    stream.reserve(whole ? nnz_all : nnz_all / 2);
    for (uint32_t c = 0; c < num_cols; c++) {
        indptr[c] = (uint32_t)stream.size();
        for (uint64_t i = h_indptr[c]; i < h_indptr[c + 1]; i++) {
            uint32_t r = h_indices[i];
            if (r >= num_rows)
                return gl::set_error(GL_ERR_INVALID_ARG, "gl_spmspv_plan_create: row index %u out of range (num_rows %u)", r, num_rows);
            if (r >= row_begin && r < row_end) stream.push_back(make_uint2(r, __builtin_bit_cast(uint32_t, h_data[i])));
        }
    }
    indptr[num_cols] = (uint32_t)stream.size();
    }
    const uint64_t kept = on_device ? (uint64_t)indptr[num_cols] : (uint64_t)stream.size();
    uint32_t max_col_len = 0;
    for (uint32_t c = 0; c < num_cols; c++) max_col_len = std::max(max_col_len, indptr[c + 1] - indptr[c]);

    gl_spmspv_plan p = new gl_spmspv_plan_s();
    p->max_col_len = max_col_len;
    p->num_rows = num_rows;
    p->num_cols = num_cols;
    p->row_begin = row_begin;
    p->row_end = row_end;
    p->nnz = kept;
    p->d_indptr = dev_indptr;   // (null unless the device built them)
    p->d_stream = dev_stream;
    const uint32_t nrows = row_end - row_begin;
    p->tiles = gl::choose_tiles(nrows, (uint32_t)gl::ctx().num_cus);
    p->binned = p->tiles.count <= gl::kBinMaxTiles;
    p->fold_tickets = p->tiles.count > (uint32_t)gl::ctx().num_cus;
    std::vector<uint4> long_chunks;   // (gl_bfs_bits_push_step: its own, shorter chunks)
    for (uint32_t c = 0; c < num_cols; c++) {
        const uint32_t d = indptr[c + 1] - indptr[c];
        if (d < gl::kBfsChunk) continue;
        for (uint32_t k = 0; k < d; k += gl::kBfsChunk) long_chunks.push_back(make_uint4(c, indptr[c] + k, std::min(gl::kBfsChunk, d - k), 0u));
    }
    p->n_long_chunks = (uint32_t)long_chunks.size();
    auto fail = [&](hipError_t e) {
        gl_spmspv_plan_destroy(p);
        return gl::set_error(GL_ERR_HIP, "gl_spmspv_plan_create: %s", hipGetErrorString(e));
    };
    hipError_t e;
    size_t b_indptr = indptr.size() * sizeof(uint32_t), b_stream = kept * sizeof(uint2);
    size_t b_acc = (size_t)(nrows ? nrows : 1) * sizeof(float);
    const uint32_t ntiles = p->tiles.count;
    size_t b_tiles = (size_t)(ntiles + 1u) * sizeof(uint32_t);
    size_t b_queue = 2u * (size_t)gl::kBinMaxSlices * sizeof(unsigned long long);
    size_t b_bins = (p->binned && kept) ? kept * sizeof(uint2) : 16;
    if (!on_device && (e = hipMalloc((void **)&p->d_indptr, b_indptr)) != hipSuccess) return fail(e);
    if (!on_device && (e = hipMalloc((void **)&p->d_stream, b_stream ? b_stream : 16)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&p->d_acc, b_acc)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&p->d_slices, b_queue)) != hipSuccess) return fail(e);
    if ((e = hipMemset(p->d_slices, 0, b_queue)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&p->d_sync, gl::kSyncWords * sizeof(uint32_t))) != hipSuccess) return fail(e);
    if ((e = hipMemset(p->d_sync, 0, gl::kSyncWords * sizeof(uint32_t))) != hipSuccess) return fail(e);
    {
        const uint32_t one = 1u;   // the bin kernel's generation: tags of the (all-zero) slice words start at 1
        if ((e = hipMemcpy(p->d_sync + gl::kSyncGen, &one, sizeof(one), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    }
    if ((e = hipMalloc((void **)&p->d_bins, b_bins)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&p->d_bin_base, b_tiles)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&p->d_cursor, b_tiles)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&p->d_state, b_tiles)) != hipSuccess) return fail(e);
    if ((e = hipMemset(p->d_bin_base, 0, b_tiles)) != hipSuccess) return fail(e);
    if ((e = hipMemset(p->d_cursor, 0, b_tiles)) != hipSuccess) return fail(e);
    if ((e = hipMemset(p->d_state, 0, b_tiles)) != hipSuccess) return fail(e);
    if ((e = hipHostMalloc((void **)&p->h_rec, 64, hipHostMallocDefault)) != hipSuccess) return fail(e);
    *p->h_rec = 0ull;
    if ((e = hipMalloc((void **)&p->d_long_chunks, (long_chunks.size() + 1u) * sizeof(uint4))) != hipSuccess) return fail(e);
    if (!long_chunks.empty() &&
        (e = hipMemcpy(p->d_long_chunks, long_chunks.data(), long_chunks.size() * sizeof(uint4), hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e);
    if ((e = hipMalloc((void **)&p->d_bfs_acc, gl::kBfsAccSlots * 128u)) != hipSuccess) return fail(e);
    if ((e = hipMemset(p->d_bfs_acc, 0, gl::kBfsAccSlots * 128u)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&p->d_mode, 16)) != hipSuccess) return fail(e);
    if ((e = hipMemset(p->d_mode, 0, 16)) != hipSuccess) return fail(e);
    if (!on_device && (e = hipMemcpy(p->d_indptr, indptr.data(), b_indptr, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if (!on_device && b_stream && (e = hipMemcpy(p->d_stream, stream.data(), b_stream, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(e);   // the memsets above ran on the null stream
    if (p->binned && kept) {
        // the bins' capacities: non-zeros per row tile, counted on the device from the stream just built (d_cursor serves as the
        // counter and is cleared again)
        gl::spmspv_tile_count_kernel<<<std::min<uint32_t>(gl::cdiv(kept, 8192), (uint32_t)gl::ctx().num_cus * 2u), 1024>>>(
            p->d_stream, kept, row_begin, p->tiles, p->d_cursor);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e);
        std::vector<uint32_t> cap(ntiles + 1u, 0u);
        if ((e = hipMemcpy(cap.data(), p->d_cursor, (size_t)ntiles * sizeof(uint32_t), hipMemcpyDeviceToHost)) != hipSuccess) return fail(e);
        uint64_t run = 0;
        for (uint32_t t = 0; t <= ntiles; t++) {
            const uint32_t c = t < ntiles ? cap[t] : 0u;
            cap[t] = (uint32_t)run;
            run += c;
        }
        if (run != kept) {
            gl_spmspv_plan_destroy(p);
            return gl::set_error(GL_ERR_HIP, "gl_spmspv_plan_create: tile histogram counted %llu of %llu non-zeros", (unsigned long long)run, (unsigned long long)kept);
        }
        if ((e = hipMemcpy(p->d_bin_base, cap.data(), b_tiles, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
        if ((e = hipMemset(p->d_cursor, 0, b_tiles)) != hipSuccess) return fail(e);
        if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(e);
    }
    p->device_bytes = b_indptr + b_stream + b_acc + 3u * b_tiles + b_queue + b_bins + long_chunks.size() * sizeof(uint4);
    gl::live_spmspv_plans().push_back(p);
    *plan = p;
    return GL_OK;
}

int gl_spmspv_plan_destroy(gl_spmspv_plan p) {
    if (!p) return GL_OK;
    std::vector<gl_spmspv_plan> &live = gl::live_spmspv_plans();
    live.erase(std::remove(live.begin(), live.end(), p), live.end());
    (void)hipFree(p->d_indptr);
    (void)hipFree(p->d_stream);
    (void)hipFree(p->d_acc);
    (void)hipFree(p->d_slices);
    (void)hipFree(p->d_sync);
    (void)hipFree(p->d_bins);
    (void)hipFree(p->d_bin_base);
    (void)hipFree(p->d_cursor);
    (void)hipFree(p->d_state);
    if (p->h_rec) (void)hipHostFree(p->h_rec);
    (void)hipFree(p->d_mode);
    (void)hipFree(p->d_long_chunks);
    (void)hipFree(p->d_bfs_acc);
    (void)hipFree(p->d_xdense);
    delete p;
    return GL_OK;
}

int gl_spmspv_plan_info(gl_spmspv_plan p, uint64_t *nnz, uint64_t *device_bytes) {
    GL_ARG(p != nullptr);
    if (nnz) *nnz = p->nnz;
    if (device_bytes) *device_bytes = p->device_bytes;
    return GL_OK;
}

int gl_spmspv_run(gl_spmspv_plan p, const gl_idx_val *d_vector, const float *d_mask, gl_idx_val *d_result,
                  int op, float zero, int mask_type) {
    return gl_spmspv_run_assign(p, d_vector, d_mask, d_result, op, zero, mask_type, nullptr, 0.0f);
}

static int spmspv_run_impl(gl_spmspv_plan p, const gl_idx_val *d_vector, const float *d_mask, gl_idx_val *d_result,
                           int op, float zero, int mask_type, float *d_inout, float val, int val_type);

int gl_spmspv_run_assign(gl_spmspv_plan p, const gl_idx_val *d_vector, const float *d_mask, gl_idx_val *d_result,
                         int op, float zero, int mask_type, float *d_inout, float val) {
    GL_TRACE();
    return spmspv_run_impl(p, d_vector, d_mask, d_result, op, zero, mask_type, d_inout, val, GL_VAL_FLOAT);
}

/* the sparse elements of the integer value types are {uint32 index; uint32 value bits}: same size and layout */
int gl_spmspv_run_typed(gl_spmspv_plan p, const void *d_vector, const void *d_mask, void *d_result, int op, uint32_t zero_bits,
                        int mask_type, int val_type) {
    if (val_type != GL_VAL_FLOAT && val_type != GL_VAL_UNSIGNED && val_type != GL_VAL_UFIXED_32_8)
        return gl::set_error(GL_ERR_INVALID_ARG, "gl_spmspv_run_typed: unknown value type %d", val_type);
    return spmspv_run_impl(p, (const gl_idx_val *)d_vector, (const float *)d_mask, (gl_idx_val *)d_result, op,
                           __builtin_bit_cast(float, zero_bits), mask_type, nullptr, 0.0f, val_type);
}

static int spmspv_run_impl(gl_spmspv_plan p, const gl_idx_val *d_vector, const float *d_mask, gl_idx_val *d_result,
                           int op, float zero, int mask_type, float *d_inout, float val, int val_type) {
    GL_REQUIRE_INIT();
    GL_ARG(p != nullptr && d_vector != nullptr && d_result != nullptr);
    GL_ARG(mask_type == GL_NOMASK || d_mask != nullptr);
    GL_ARG(op == GL_OP_MULADD || op == GL_OP_ANDOR || op == GL_OP_ADDMIN);
    GL_ARG(mask_type == GL_NOMASK || mask_type == GL_MASK_WRITETOZERO || mask_type == GL_MASK_WRITETOONE);
    hipStream_t s = gl::ctx().stream;
    const uint32_t nrows = p->row_end - p->row_begin;

    // accumulator invariant: all entries == zero on entry (bitwise compare so -0/NaN refill too)
    if (!p->acc_valid || memcmp(&p->acc_fill, &zero, sizeof(float)) != 0) {
        int rc = gl_buf_fill_f32(p->d_acc, zero, nrows);
        if (rc != GL_OK) return rc;
        p->acc_fill = zero;
        p->acc_valid = true;
    }

    // a caller that expects a tiny vector (gl_spmspv_plan_hint_tiny): the whole run is one launch of one workgroup
    const bool tiny = p->tiny_hint && val_type == GL_VAL_FLOAT && nrows > 0 && gl::debug_knob("spmspv_tiny", 1) != 0;
    p->tiny_hint = false;
    const uint64_t work_hint = gl::debug_knob("spmspv_work_hint", 1) != 0 ? p->work_hint : ~0ull;
    p->work_hint = ~0ull;
    const uint32_t nnz_hint = p->nnz_hint;
    p->nnz_hint = ~0u;
    p->rec_pending = false;
    if (tiny) {
        p->frontier_hint = ~0ull;
        p->last_decided_on_device = false;
        gl::TinyArgs t;
        t.indptr = p->d_indptr;
        t.stream = p->d_stream;
        t.vec = d_vector;
        t.acc = p->d_acc;
        t.row_begin = p->row_begin;
        t.nrows = nrows;
        t.num_cols = p->num_cols;
        t.out = d_result;
        t.head_val = zero;
        t.zero_bits = __builtin_bit_cast(uint32_t, zero);
        t.bucket_shift = 0u;
        while (((nrows - 1u) >> t.bucket_shift) >= 2048u) t.bucket_shift++;
        {   // (a run that is not being recorded into a graph reports its completion to the host, as below)
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(s, &cap);
            const bool report = cap == hipStreamCaptureStatusNone && p->h_rec != nullptr;
            t.host_rec = report ? p->h_rec : nullptr;
            t.seq = report ? ++p->seq : 0u;
            p->rec_pending = report;
            p->rec_epoch = gl::graph_launches();
        }
        switch (op) {
            case GL_OP_MULADD: return gl::launch_tiny_mask<GL_OP_MULADD>(mask_type, t, d_mask, zero, d_inout, val, s);
            case GL_OP_ANDOR: return gl::launch_tiny_mask<GL_OP_ANDOR>(mask_type, t, d_mask, zero, d_inout, val, s);
            default: return gl::launch_tiny_mask<GL_OP_ADDMIN>(mask_type, t, d_mask, zero, d_inout, val, s);
        }
    }

    // (||,&&) with an attached boolean SpMV plan: decide on the device which way this run goes
    const long div = gl::env_long("GRAPHLILY_SPMSPV_PULL_DIV", 8);      // (0: never row-wise)
    const uint64_t threshold = div > 0 ? p->nnz / (uint64_t)div : 0ull;
    // which attached plan can stand in for the scatter: (||,&&) and (+,x) need zero == 0 (the accumulator starts
    // at it); (min,+) needs zero <= FLOAT_INF -- the scatter's products saturate there, the SpMV's do not, and the
    // final min with zero hides the difference -- and a plan that was not restricted to other semirings
    gl_spmv_plan pull_plan = nullptr;
    if (op == GL_OP_ANDOR && zero == 0.0f) pull_plan = p->pull;
    else if (op == GL_OP_MULADD && zero == 0.0f && p->pull_arith && !(p->pull_arith->flags & GL_PLAN_NO_MULADD)) pull_plan = p->pull_arith;
    else if (op == GL_OP_ADDMIN && zero <= gl::kFloatInf) pull_plan = p->pull_arith;
    bool may_pull = pull_plan != nullptr && nrows > 0 && val_type == GL_VAL_FLOAT && div > 0;
    // a caller that knows how many entries the vector holds (gl_spmspv_plan_hint) spares tiny frontiers the
    // decision kernels: they cannot reach the threshold whatever their columns are
    if (may_pull && p->frontier_hint != ~0ull && p->frontier_hint * (uint64_t)p->max_col_len <= threshold) may_pull = false;
    p->frontier_hint = ~0ull;
    // ... and one that knows the work itself (gl_spmspv_plan_hint_work: a module that uploaded the vector from the host) takes
    // the decision of spmspv_work_kernel here when it is "scatter": no decision kernel and none of the row-wise kernels that
    // would only find out that they have nothing to do (five dependent launches, ~22 us of a 60 us call).  A stale hint costs
    // time, never results: scattering is correct for any vector.
    if (may_pull && work_hint != ~0ull && work_hint <= threshold) may_pull = false;
    if (may_pull) {
        // few blocks: each ends with one atomic on the same ticket word
        uint32_t wgrid = std::min<uint32_t>(gl::cdiv(p->num_cols, 256), 64u);
        gl::spmspv_work_kernel<<<wgrid ? wgrid : 1u, 256, 0, s>>>(d_vector, p->d_indptr, p->num_cols, p->d_mode, threshold);
        GL_LAUNCH_CHECK();
    }
    p->last_decided_on_device = may_pull;   // gl_spmspv_last_direction: else the run scattered, nothing to read back

    gl::BinArgs a;
    a.mode = may_pull ? p->d_mode : nullptr;
    a.indptr = p->d_indptr;
    a.stream = p->d_stream;
    a.vec = d_vector;
    a.bins = p->d_bins;
    a.bin_base = p->d_bin_base;
    a.cursor = p->d_cursor;
    a.acc = p->d_acc;
    a.sync = p->d_sync;
    a.slices = p->d_slices;
    a.tiles = p->tiles;
    a.binned = p->binned ? 1u : 0u;
    a.row_begin = p->row_begin;
    a.num_cols = p->num_cols;
    a.max_col_len = p->max_col_len;
    a.by_entries = p->max_col_len <= (uint32_t)gl::debug_knob("spmspv_by_entries_maxcol", gl::kBinByEntriesMaxCol) ? 1u : 0u;
    // at most one workgroup per compute unit (the kernel's rendezvous waits for every workgroup of the grid: all must be
    // resident), fewer when the caller has said how short the vector is
    uint32_t grid = (uint32_t)gl::ctx().num_cus;
    if (work_hint != ~0ull) grid = (uint32_t)std::min<uint64_t>(grid, std::max<uint64_t>((work_hint + 2047u) / 2048u, 1u));
    else if (nnz_hint != ~0u && (uint64_t)nnz_hint * p->max_col_len < 2048ull * grid)
        grid = (uint32_t)std::max<uint64_t>(((uint64_t)nnz_hint * p->max_col_len + 2047u) / 2048u, 1u);
    if (grid == 0) grid = 1;
    int rc = GL_OK;
    if (nrows > 0) {
        switch (op + 3 * val_type) {
            case GL_OP_MULADD: gl::spmspv_bin_kernel<GL_OP_MULADD><<<grid, gl::kBinThreads, 0, s>>>(a); break;
            case GL_OP_ANDOR: gl::spmspv_bin_kernel<GL_OP_ANDOR><<<grid, gl::kBinThreads, 0, s>>>(a); break;
            case GL_OP_ADDMIN: gl::spmspv_bin_kernel<GL_OP_ADDMIN><<<grid, gl::kBinThreads, 0, s>>>(a); break;
            case gl::kOpU32MulAdd: gl::spmspv_bin_kernel<gl::kOpU32MulAdd><<<grid, gl::kBinThreads, 0, s>>>(a); break;
            case gl::kOpU32AndOr: gl::spmspv_bin_kernel<gl::kOpU32AndOr><<<grid, gl::kBinThreads, 0, s>>>(a); break;
            case gl::kOpU32AddMin: gl::spmspv_bin_kernel<gl::kOpU32AddMin><<<grid, gl::kBinThreads, 0, s>>>(a); break;
            case gl::kOpFixMulAdd: gl::spmspv_bin_kernel<gl::kOpFixMulAdd><<<grid, gl::kBinThreads, 0, s>>>(a); break;
            case gl::kOpFixAndOr: gl::spmspv_bin_kernel<gl::kOpFixAndOr><<<grid, gl::kBinThreads, 0, s>>>(a); break;
            case gl::kOpFixAddMin: gl::spmspv_bin_kernel<gl::kOpFixAddMin><<<grid, gl::kBinThreads, 0, s>>>(a); break;
            default: return gl::set_error(GL_ERR_UNSUPPORTED, "gl_spmspv_run: semiring op %d is not offered for value type %d", op, val_type);
        }
        GL_LAUNCH_CHECK();
    }
    if (may_pull && op != GL_OP_ANDOR) {
        // row-wise (+,x) / (min,+): frontier -> dense x (0 / +inf elsewhere) -> SpMV on the attached general /
        // pattern plan into the accumulator (same zero, no mask: the fold applies the mask); its kernels
        // return at once on a run that bins
        uint32_t bgrid = std::min<uint32_t>(gl::cdiv(p->num_cols, 256), (uint32_t)gl::ctx().num_cus * 8u);
        gl::spmspv_clear_dense_kernel<<<bgrid ? bgrid : 1u, 256, 0, s>>>(reinterpret_cast<float4 *>(p->d_xdense), gl::cdiv(p->num_cols, 4),
                                                                        op == GL_OP_MULADD ? 0.0f : __builtin_inff(), p->d_mode);
        GL_LAUNCH_CHECK();
        gl::spmspv_frontier_dense_kernel<<<bgrid ? bgrid : 1u, 256, 0, s>>>(d_vector, p->num_cols, p->d_xdense, p->d_mode);
        GL_LAUNCH_CHECK();
        rc = gl::spmv_run_general(p->pull_arith, p->d_xdense, nullptr, p->d_acc - p->row_begin, op, zero, GL_NOMASK, p->d_mode);
        if (rc != GL_OK) return rc;
    } else if (may_pull) {
        // row-wise: frontier -> bit vector -> boolean SpMV into the (all-zero) accumulator; both kernels return
        // at once when the run bins
        uint32_t *bits = gl::bool_plan_xbits(p->pull);
        GL_HIP(hipMemsetAsync(bits, 0, gl::bool_plan_xbits_bytes(p->pull), s));
        uint32_t bgrid = std::min<uint32_t>(gl::cdiv(p->num_cols, 256), (uint32_t)gl::ctx().num_cus * 8u);
        gl::spmspv_frontier_bits_kernel<<<bgrid ? bgrid : 1u, 256, 0, s>>>(d_vector, p->num_cols, bits, p->d_mode);
        GL_LAUNCH_CHECK();
        rc = gl::bool_plan_run_bits(p->pull, p->d_acc - p->row_begin, p->d_mode, s);
        if (rc != GL_OK) return rc;
    }

    // ---- fold: one workgroup per row tile
    gl::FoldArgs f;
    f.bins = p->d_bins;
    f.bin_base = p->d_bin_base;
    f.cursor = p->d_cursor;
    f.acc = p->d_acc;
    f.mask = d_mask;
    f.mask_type = mask_type;
    f.nrows = nrows;
    f.row_begin = p->row_begin;
    f.zero = zero;
    f.tiles = p->tiles;
    f.out = d_result;
    f.head_val = zero;
    f.assign = d_inout;
    f.assign_val = val;
    f.state = p->d_state;
    f.sync = p->d_sync;
    f.tickets = p->fold_tickets ? 1u : 0u;
    f.merge_all = p->binned ? 0u : 1u;
    f.mode = may_pull ? p->d_mode : nullptr;
    f.bin_vec = nrows > 0 ? d_vector : nullptr;
    f.bin_grid = grid;
    // a run that is not being recorded into a graph reports its completion to the host (gl_spmspv_wait)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    const bool report = cap == hipStreamCaptureStatusNone && p->h_rec != nullptr;
    f.host_rec = report ? p->h_rec : nullptr;
    f.seq = report ? ++p->seq : 0u;
    p->rec_pending = report;
    p->rec_epoch = gl::graph_launches();
    // test hook (GRAPHLILY_DEBUG spmspv_inject_timeout=1): mark this run as one whose rendezvous timed out, the way a poll that gave
    // up does, so that tests can drive the failure path -- reporting runs AND runs recorded into a graph -- without a real timeout
    if (gl::debug_knob("spmspv_inject_timeout", 0) != 0) {
        gl::spmspv_inject_timeout_kernel<<<1, 1, 0, s>>>(p->d_sync);
        GL_LAUNCH_CHECK();
    }
    return gl::launch_fold(op + 3 * val_type, f, s);
}

// the record of the plan's last run has arrived: hand its count out ONCE (a later call, with other work enqueued since, reads the
// head element instead), or report the run's failed rendezvous (gl_spmspv_bin.h kSyncErr)
static int spmspv_take_record(gl_spmspv_plan p, unsigned long long v, uint32_t *nnz) {
    p->rec_pending = false;
    if ((uint32_t)v == 0xffffffffu)
        return gl::set_error(GL_ERR_HIP, "gl_spmspv_run: a workgroup rendezvous timed out (another kernel kept workgroups of the run from "
                             "being resident); the result list was emptied");
    if (nnz) *nnz = (uint32_t)v;
    return GL_OK;
}

int gl_spmspv_wait(gl_spmspv_plan p, uint32_t *nnz) {
    GL_TRACE();
    GL_REQUIRE_INIT();
    GL_ARG(p != nullptr);
    if (p->rec_pending && p->rec_epoch != gl::graph_launches()) p->rec_pending = false;   // a graph replay may have rerun this plan
    if (p->rec_pending) {
        // the fold's last workgroup stores seq << 32 | count to page-locked memory once every result has been written: the host
        // sees it ~5 us before hipStreamSynchronize returns (profiles/r03_ubench_sync.txt)
        volatile unsigned long long *rec = p->h_rec;
        const unsigned long long want = (unsigned long long)p->seq;
        for (uint64_t spins = 0;; spins++) {
            const unsigned long long v = *rec;
            if ((v >> 32) == want) return spmspv_take_record(p, v, nnz);
            if (spins > (1ull << 16) && hipStreamQuery(gl::ctx().stream) != hipErrorNotReady) break;   // the stream is idle (or failed)
            __builtin_ia32_pause();
        }
        GL_HIP(hipStreamSynchronize(gl::ctx().stream));
        const unsigned long long v = *rec;
        if ((v >> 32) == want) return spmspv_take_record(p, v, nnz);
        return gl::set_error(GL_ERR_HIP, "gl_spmspv_wait: the run finished without its completion record (sequence %u, found %u)", p->seq,
                             (uint32_t)(v >> 32));
    }
    GL_HIP(hipStreamSynchronize(gl::ctx().stream));
    if (nnz) *nnz = 0xffffffffu;   // no record: read the head element (gl_sparse_nnz)
    return GL_OK;
}

int gl_spmspv_failed_runs(gl_spmspv_plan p, uint32_t *count) {
    GL_REQUIRE_INIT();
    GL_ARG(p != nullptr && count != nullptr);
    hipStream_t s = gl::ctx().stream;
    GL_HIP(gl::d2h_word_sync(count, p->d_sync + gl::kSyncFailedRuns, s));
    return GL_OK;
}

int gl_spmspv_plan_attach_pull(gl_spmspv_plan p, gl_spmv_plan pull) {
    GL_ARG(p != nullptr);
    if (pull == nullptr) {
        p->pull = p->pull_arith = nullptr;
        return GL_OK;
    }
    if (pull->num_rows != p->num_rows || pull->num_cols != p->num_cols || pull->row_begin != p->row_begin ||
        pull->row_end != p->row_end)
        return gl::set_error(GL_ERR_INVALID_ARG, "gl_spmspv_plan_attach_pull: the two plans hold different matrices or row shards");
    if (pull->boolean) {
        p->pull = pull;
        return GL_OK;
    }
    if (!p->d_xdense) {
        GL_HIP(hipMalloc((void **)&p->d_xdense, ((size_t)p->num_cols + 4u) * sizeof(float)));   // whole float4s
        p->device_bytes += (size_t)p->num_cols * sizeof(float);
    }
    p->pull_arith = pull;
    return GL_OK;
}

int gl_spmspv_plan_hint_work(gl_spmspv_plan p, uint32_t vector_nnz, uint64_t work, uint32_t longest_column) {
    GL_ARG(p != nullptr);
    p->tiny_hint = vector_nnz <= gl::kTinyVec && work <= gl::kTinyWork;
    p->work_hint = work;
    p->longest_hint = longest_column;
    p->nnz_hint = vector_nnz;
    return GL_OK;
}

int gl_spmspv_plan_hint(gl_spmspv_plan p, uint32_t vector_nnz_upper_bound) {
    GL_ARG(p != nullptr);
    p->frontier_hint = vector_nnz_upper_bound;
    p->nnz_hint = vector_nnz_upper_bound;
    return GL_OK;
}

int gl_spmspv_last_direction(gl_spmspv_plan p, int *row_wise) {
    GL_REQUIRE_INIT();
    GL_ARG(p != nullptr && row_wise != nullptr);
    uint32_t m = 0;
    if (p->last_decided_on_device) {
        hipStream_t s = gl::ctx().stream;
        GL_HIP(gl::d2h_word_sync(&m, p->d_mode, s));
    }
    *row_wise = ((p->pull != nullptr || p->pull_arith != nullptr) && m != 0u) ? 1 : 0;
    return GL_OK;
}

int gl_bfs_bits_push_step(gl_spmspv_plan p, gl_spmv_plan rows, const uint32_t *d_bits_in, uint32_t *d_bits_out, uint32_t *d_bits_spare,
                          uint32_t bits_words, float *d_distance, float level, uint32_t *d_ctl, uint32_t slot, float threshold,
                          int may_continue) {
    GL_REQUIRE_INIT();
    GL_ARG(p != nullptr && d_bits_in != nullptr && d_bits_out != nullptr);
    GL_ARG(d_distance != nullptr && d_ctl != nullptr && slot >= 1u && ((uintptr_t)d_ctl & 7u) == 0);
    GL_ARG(d_bits_in != d_bits_out && d_bits_in != d_bits_spare && d_bits_out != d_bits_spare);
    GL_ARG((uint64_t)bits_words * 32u >= p->num_cols && (uint64_t)bits_words * 32u >= p->num_rows);
    // the bottom-up branch writes the next frontier as whole 64-bit words
    GL_ARG((bits_words & 1u) == 0 && (((uintptr_t)d_bits_in | (uintptr_t)d_bits_out) & 7u) == 0);
    const bool deferred = (may_continue & GL_BFS_DEFERRED) != 0;
    if ((p->row_begin != 0 || p->row_end != p->num_rows) && !deferred)
        return gl::set_error(GL_ERR_UNSUPPORTED, "gl_bfs_bits_push_step: a row shard's frontier counts are partial -- pass GL_BFS_DEFERRED "
                             "in may_continue and take the slot's decisions with gl_bfs_bits_decide after the all-gather");
    GL_ARG((p->row_begin & 63u) == 0u && (p->row_end == p->num_rows || (p->row_end & 63u) == 0u));
    gl::BfsPushArgs a;
    a.indptr = p->d_indptr;
    a.stream = p->d_stream;
    a.chunks = p->d_long_chunks;
    a.nchunks = p->n_long_chunks;
    a.num_cols = p->num_cols;
    a.bits_in = d_bits_in;
    a.bits_out = d_bits_out;
    a.bits_spare = d_bits_spare;
    a.words = bits_words;
    a.col_words = gl::cdiv(p->num_cols, 32);
    a.dist = d_distance;
    a.level = level;
    a.acc = p->d_bfs_acc;
    const bool have_rows = rows != nullptr && rows->d_csr_indptr != nullptr && rows->num_rows == p->num_rows &&
                           rows->num_cols == p->num_cols && rows->row_begin == p->row_begin && rows->row_end == p->row_end;
    p->bfs_rows_plan = have_rows ? rows : nullptr;
    a.row_ptr = have_rows ? rows->d_csr_indptr : nullptr;
    a.row_idx = have_rows ? rows->d_csr_indices : nullptr;
    a.num_rows = p->num_rows;
    a.row_begin = p->row_begin;
    a.row_end = p->row_end;
    a.nz_base = have_rows ? rows->csr_nz_base : 0u;
    a.deferred = deferred;
    a.c.ctl = d_ctl;
    a.c.slot = slot;
    a.c.n = p->num_rows ? p->num_rows : 1u;
    a.c.may_continue = (uint32_t)may_continue & 3u;
    a.c.threshold = threshold;
    a.c.back_threshold = 0.0f;
    a.c.heavy = gl::spmspv_heavy_work(p);
    uint32_t grid = std::min<uint32_t>((uint32_t)gl::ctx().num_cus * 8u, std::max<uint32_t>(gl::cdiv(a.col_words, 8), 1u));
    gl::bfs_push_bits_kernel<<<grid, 256, 0, gl::ctx().stream>>>(a);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// One slot of the row-sharded schedule in one launch (gl_bfs_shard.h).  finish: only the decision of slot `slot - 1` = the
// last one, with the final state stored into d_ctl itself.
static int bfs_bits_shard_launch(gl_spmspv_plan p, gl_spmv_plan rows, const uint32_t *d_bits_in, uint32_t *d_bits_out, uint32_t bits_words,
                                 float *d_distance, float level, uint32_t *d_ctl, uint32_t *d_tally, const uint32_t *d_tally_in,
                                 uint32_t slot, int rank, int world, const uint32_t *d_col_len, uint64_t nnz_global, float threshold,
                                 int may_continue_prev, float back_threshold, bool finish) {
    GL_REQUIRE_INIT();
    GL_ARG(p != nullptr && rows != nullptr && d_ctl != nullptr && d_tally != nullptr && slot >= 1u);
    GL_ARG(world >= 1 && rank >= 0 && rank < world && (((uintptr_t)d_ctl | (uintptr_t)d_tally) & 15u) == 0);
    if (!rows->boolean)
        return gl::set_error(GL_ERR_UNSUPPORTED, "gl_bfs_bits_shard_step: the row plan does not hold the GL_PLAN_BOOLEAN layout");
    GL_ARG(rows->num_rows == p->num_rows && rows->num_cols == p->num_cols && rows->row_begin == p->row_begin && rows->row_end == p->row_end);
    if (!d_tally_in) d_tally_in = d_tally;
    gl::BfsPushArgs a;
    a.indptr = p->d_indptr;
    a.stream = p->d_stream;
    a.chunks = p->d_long_chunks;
    a.nchunks = p->n_long_chunks;
    a.num_cols = p->num_cols;
    a.bits_in = d_bits_in;
    a.bits_out = d_bits_out;
    a.bits_spare = nullptr;
    a.words = bits_words;
    a.col_words = gl::cdiv(p->num_cols, 32);
    a.dist = d_distance;
    a.level = level;
    a.acc = nullptr;
    const bool have_rows = rows->d_csr_indptr != nullptr;
    p->bfs_rows_plan = have_rows ? rows : nullptr;
    a.row_ptr = have_rows ? rows->d_csr_indptr : nullptr;
    a.row_idx = have_rows ? rows->d_csr_indices : nullptr;
    a.num_rows = p->num_rows;
    a.row_begin = p->row_begin;
    a.row_end = p->row_end;
    a.nz_base = have_rows ? rows->csr_nz_base : 0u;
    a.deferred = false;
    a.col_len = d_col_len;
    gl::BfsShardArgs sa;
    // the state after slot k lives in d_ctl for even k and in the head of d_tally for odd k: the launch of slot s reads the
    // state after s - 2 (after s - 1 = 0 in slot 1) and stores the one after s - 1 into the buffer nobody reads meanwhile
    uint32_t *buf[2] = {d_ctl, d_tally};
    sa.state_in = slot == 1u ? d_ctl : buf[slot & 1u];
    sa.state_out = slot == 1u ? nullptr : (finish ? d_ctl : buf[(slot - 1u) & 1u]);
    sa.records = d_ctl;
    const size_t per_slot = (size_t)world * gl::kTallyRankWords;
    sa.tally_prev = slot == 1u ? nullptr : d_tally_in + gl::kTallyHeadWords + (size_t)(slot - 2u) * per_slot;
    sa.lines_prev = (uint32_t)world * gl::kTallyLines;
    sa.tally_mine = d_tally + gl::kTallyHeadWords + (size_t)(slot - 1u) * per_slot + (size_t)rank * gl::kTallyRankWords;
    sa.slot = slot;
    sa.finish = finish ? 1u : 0u;
    sa.pull_units = 0;
    // (one workgroup per compute unit fits next to the pull's LDS tile: a larger grid would run in rounds)
    // bits per lane of the scattering push: halve the strips until they cover the wavefronts of a one-workgroup-per-CU grid
    {
        const uint64_t waves = (uint64_t)gl::ctx().num_cus * (gl::kThreads / 64u), bits = (uint64_t)a.col_words * 32u;
        uint32_t bpl = 32u;
        while (bpl > 1u && 2u * gl::cdiv(bits, 64u * bpl) <= waves) bpl >>= 1;
        a.bpl = bpl;
        const uint32_t nstrips = std::max<uint32_t>(gl::cdiv(bits, 64u * bpl), 1u);
        sa.push_blocks = std::min<uint32_t>((uint32_t)gl::ctx().num_cus, gl::cdiv(nstrips, gl::kThreads / 64u));
    }
    gl::BfsBitsCtl &c = sa.prev;
    c.ctl = nullptr;
    c.slot = slot - 1u;
    c.n = p->num_rows ? p->num_rows : 1u;
    c.may_continue = (uint32_t)may_continue_prev & 3u;
    c.threshold = threshold;
    c.back_threshold = back_threshold;
    // the GLOBAL matrix decides, as in gl_bfs_bits_decide
    const long hdiv = gl::debug_knob("bfs_heavy_div", 128), bdiv = gl::debug_knob("bfs_bu_div", 3);
    c.heavy = hdiv > 0 ? nnz_global / (unsigned long long)hdiv : ~0ull;
    c.nnz_rows = nnz_global;
    c.bu_limit = (have_rows && bdiv > 0) ? nnz_global / (unsigned long long)bdiv : 0ull;
    if (!finish) {
        GL_ARG(d_bits_in != nullptr && d_bits_out != nullptr && d_distance != nullptr && d_col_len != nullptr && d_bits_in != d_bits_out);
        GL_ARG((uint64_t)bits_words * 32u >= p->num_cols && (uint64_t)bits_words * 32u >= p->num_rows);
        GL_ARG((bits_words & 1u) == 0 && (((uintptr_t)d_bits_in | (uintptr_t)d_bits_out) & 15u) == 0);
    }
    return gl::bool_plan_bfs_shard_step(rows, a, sa, gl::ctx().stream);
}

int gl_bfs_bits_shard_step(gl_spmspv_plan p, gl_spmv_plan rows, const uint32_t *d_bits_in, uint32_t *d_bits_out, uint32_t bits_words,
                           float *d_distance, float level, uint32_t *d_ctl, uint32_t *d_tally, const uint32_t *d_tally_in, uint32_t slot,
                           int rank, int world_size, const uint32_t *d_col_len, uint64_t nnz_global, float threshold,
                           int may_continue_prev, float back_threshold) {
    return bfs_bits_shard_launch(p, rows, d_bits_in, d_bits_out, bits_words, d_distance, level, d_ctl, d_tally, d_tally_in, slot, rank,
                                 world_size, d_col_len, nnz_global, threshold, may_continue_prev, back_threshold, false);
}

int gl_bfs_bits_shard_finish(gl_spmspv_plan p, gl_spmv_plan rows, uint32_t *d_ctl, uint32_t *d_tally, const uint32_t *d_tally_in,
                             uint32_t last_slot, int rank, int world_size, uint64_t nnz_global, float threshold, int may_continue_last,
                             float back_threshold) {
    return bfs_bits_shard_launch(p, rows, nullptr, nullptr, 0u, nullptr, 0.0f, d_ctl, d_tally, d_tally_in, last_slot + 1u, rank, world_size,
                                 nullptr, nnz_global, threshold, may_continue_last, back_threshold, true);
}

int gl_sparse_nnz(const gl_idx_val *d_sparse, uint32_t *nnz) {
    GL_TRACE();
    GL_REQUIRE_INIT();
    GL_ARG(d_sparse != nullptr && nnz != nullptr);
    hipStream_t s = gl::ctx().stream;
    // a page-locked destination makes this one DMA + one wait (a pageable one goes through a staging copy)
    uint32_t *&w = gl::ctx().pinned_word;
    if (!w) GL_HIP(hipHostMalloc((void **)&w, 64, hipHostMallocDefault));
    GL_HIP(hipMemcpyAsync(w, &d_sparse->index, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GL_HIP(hipStreamSynchronize(s));
    *nnz = *w;
    return GL_OK;
}

}  // extern "C"

// gl_init loads this translation unit's code object up front (HIP defers that to the unit's first launch, which would put
// tens of milliseconds into somebody's timed call)
namespace gl {
// The bit-frontier BFS schedule applies a frontier whose columns hold more non-zeros than this row-wise (the pull step of
// the slot).  A push step retires ~11 G products/s (three dependent round trips per 64 of them), the boolean SpMV streams
// ~1.4 T entries/s: they cost the same near nnz / 128 (same-box sweep of 32 ... 256 on the six stand-ins; the list-based
// gl_spmspv_run, whose scatter also pays for a compaction, keeps its 1 / 32).
unsigned long long spmspv_heavy_work(gl_spmspv_plan p) {
    const long div = debug_knob("bfs_heavy_div", 128);
    return div > 0 ? p->nnz / (unsigned long long)div : ~0ull;
}
// ... and visits only the rows not reached yet once those hold fewer non-zeros than this (bottom-up: a thread per row with
// an early exit reads an entry ~4 x more expensively than the streaming kernel, but stops at the first hit)
unsigned long long spmspv_bottom_up_limit(gl_spmspv_plan p) {
    const long div = debug_knob("bfs_bu_div", 3);
    return div > 0 ? p->nnz / (unsigned long long)div : 0ull;
}
unsigned long long spmspv_plan_nnz(gl_spmspv_plan p) { return p->nnz; }
const void *spmspv_plan_bfs_rows(gl_spmspv_plan p) { return p->bfs_rows_plan; }
const uint32_t *spmspv_plan_indptr(gl_spmspv_plan p) { return p->d_indptr; }
uint32_t spmspv_plan_num_cols(gl_spmspv_plan p) { return p->num_cols; }
uint32_t *spmspv_plan_bfs_acc(gl_spmspv_plan p) { return p->d_bfs_acc; }
bool spmspv_plan_whole(gl_spmspv_plan p, uint32_t num_rows) { return p->row_begin == 0 && p->row_end == p->num_rows && p->num_rows == num_rows; }

int preload_spmspv() {
    hipFuncAttributes attr;
    GL_HIP(hipFuncGetAttributes(&attr, (const void *)spmspv_work_kernel));
    return GL_OK;
}
}  // namespace gl

#if defined(GL_STAMPS)
// scratch builds only: the phase stamps of the last bin / fold launches (2 x 256 x 16 words)
extern "C" int gl_debug_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gl::g_stamps), sizeof(gl::g_stamps)) == hipSuccess ? 0 : 1;
}
#endif
