// Ordered stream compaction shared by SpMSpV (dense accumulator -> sparse
// result) and the SSSP-mode sparse assign (mask list -> new frontier).
//
// The reference emits its sparse outputs in an unspecified, lane-round-robin
// order (hw/kernel_spmspv_impl.h:253-286).  This build emits them in candidate
// order (ascending row for SpMSpV, mask order for the assign), so results are
// reproducible run to run.  Per-block counts, a scan of the counts (+ head element;
// done by the last counting block for short lists, by a one-block launch for long
// ones), ordered write.
//
// A "source" functor provides:
//   __device__ uint32_t size() const;                       // number of candidates
//   __device__ bool     get(uint32_t i, gl_idx_val &out);   // candidate i kept? payload
//   __device__ void     consumed(uint32_t i);               // called once per candidate in the write pass
//   __device__ void     emitted(const gl_idx_val &item);    // called once per kept candidate in the write pass
//   __device__ void     begin_chunk(uint32_t first);        // counting pass, once per block (all threads call it)
// d_counts: cdiv(max_items, kCompactChunk) + 1 words, word 0 (the ticket) zero before the first run.
#ifndef GL_COMPACT_H_
#define GL_COMPACT_H_

#include "gl_common.h"

#include <cstdlib>

namespace gl {

constexpr uint32_t kCompactThreads = 256;
constexpr uint32_t kCompactItems = 4;
constexpr uint32_t kCompactChunk = kCompactThreads * kCompactItems;
constexpr uint32_t kCompactFuseBlocks = 128;   // lists up to 128 K candidates scan their counts in the counting launch

// Optional predicate of a launch: when `word` is set the kernels do nothing unless *word (op) value holds, op being
// GL_GATE_EQ / GL_GATE_GT / GL_GATE_LE.  Drivers that enqueue a whole push / pull schedule without host round trips
// (gl_bfs_*_gated) use it with a device-side word that holds the first pull slot.
struct Gate {
    const uint32_t *word = nullptr;
    uint32_t value = 0;
    int op = GL_GATE_EQ;
    __device__ bool closed() const {
        if (word == nullptr) return false;
        const uint32_t w = *word;
        return !(op == GL_GATE_EQ ? w == value : op == GL_GATE_GT ? w > value : w <= value);
    }
};

// Optional epilogue of the scan: the push -> pull decision of a BFS, taken where the result count is produced
// (do { push } while (iter < num_iterations && nnz / n < threshold), app/bfs.h:180-190).  ctl[0] = first pull slot
// (0xffffffff while pushing), ctl[1] = push iterations done (bit 0 of may_continue: the reference's loop condition).  The push step of slot `slot` is gated on ctl[0] > slot,
// so writing slot + 1 here does not close the gate of the write pass that follows in the same step.
struct Direction {
    uint32_t *ctl = nullptr;
    uint32_t n = 1, slot = 0;
    float threshold = 0.0f;
    uint32_t may_continue = 0;
    // bit 3 of may_continue (GL_STEP_PULL_FLAGS): the schedule keeps one word per slot at ctl[32 + s] that says "slot s pulls"
    // (gl_spmv_run_flagged / gl_ewise_add_flagged read it as their launch predicate: SSSP's pull iterations are plain SpMV
    // runs, app/sssp.h:227-242); the decision that ends the push phase sets the words of every later slot (ctl[15] = words of ctl)
    // After a pull step has handed the loop back to pushing (gl_bfs_pull_step_back: ctl[4] = its slot, ctl[7] = the
    // threshold it used) the pushes are counted apart (ctl[3]; ctl[1] stays the reference's count), may run through the
    // last iteration (bit 1 of may_continue: a slot follows) and stay for as long as the frontier is below ctl[7].
    __device__ void decide(uint32_t nnz) const {
        if (!ctl || ctl[0] != 0xffffffffu) return;
        const bool again = ctl[4] != 0xffffffffu;
        ctl[again ? 3 : 1] += 1u;
        const bool cont = again ? (may_continue & 2u) != 0u : (may_continue & 1u) != 0u;
        const float thr = again ? __uint_as_float(ctl[7]) : threshold;
        if (!(cont && ((float)nnz / (float)n < thr))) {
            ctl[0] = slot + 1u;
            if (may_continue & 8u) {
                const uint32_t cw = ctl[15];
                for (uint32_t s = slot + 1u; 32u + s < cw; s++) ctl[32u + s] = 1u;
            }
        }
    }
};

// exclusive prefix of `flag` over the 256 threads of a block, in thread order; total in *block_total
__device__ __forceinline__ uint32_t block_rank_256(bool flag, uint32_t *lds4, uint32_t *block_total) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint64_t b = __ballot(flag);
    const uint32_t in_wave = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) lds4[w] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        uint32_t c = lds4[i];
        if (i < w) before += c;
        total += c;
    }
    __syncthreads();
    *block_total = total;
    return before + in_wave;
}

// Pass 1: per-block counts into counts[1 + block].  With `fused`, the block that finishes last (ticket in counts[0],
// zero between runs) turns them into exclusive offsets in place and writes the head element out[0] = {total,
// head_val}: one launch less (~30 us per BFS push iteration on googleplus).  Only for short lists: every block ends
// with an atomic on the same word, and thousands of them (3 M rows = 3000 blocks) cost more than the launch saved
// (orkut BFS pull-push 1.15 -> 1.30 ms when this was unconditional).
template <typename Src>
__global__ __launch_bounds__(256) void compact_count_kernel(Src src, uint32_t *__restrict__ counts, gl_idx_val *__restrict__ out,
                                                            float head_val, uint32_t *__restrict__ reset_word, bool fused, Gate gate,
                                                            Direction dir) {
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t carry_s;
    __shared__ bool last_s;
    if (gate.closed()) return;
    const uint32_t n = src.size();
    const uint32_t base = blockIdx.x * kCompactChunk;
    src.begin_chunk(base);
    uint32_t c = 0;
    if (base < n) {
#pragma unroll
        for (uint32_t j = 0; j < kCompactItems; j++) {
            uint32_t i = base + j * kCompactThreads + threadIdx.x;
            gl_idx_val tmp;
            c += (i < n && src.get(i, tmp)) ? 1u : 0u;
        }
    }
    // block sum
    for (int dlt = 32; dlt >= 1; dlt >>= 1) c += __shfl_down(c, dlt);
    if ((threadIdx.x & 63u) == 0) lds4[threadIdx.x >> 6] = c;
    __syncthreads();
    if (!fused) {
        if (threadIdx.x == 0) counts[1u + blockIdx.x] = lds4[0] + lds4[1] + lds4[2] + lds4[3];
        return;
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&counts[1u + blockIdx.x], lds4[0] + lds4[1] + lds4[2] + lds4[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last_s = atomicAdd(&counts[0], 1u) == gridDim.x - 1u;   // no block waits for another: the last one does the scan
        carry_s = 0;
    }
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    const uint32_t nblocks = gridDim.x, lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += kCompactThreads) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = (i < nblocks) ? __hip_atomic_load(&counts[1u + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        uint32_t incl = v;
#pragma unroll
        for (uint32_t dlt = 1; dlt < 64; dlt <<= 1) {
            uint32_t up = __shfl_up(incl, dlt);
            if (lane >= dlt) incl += up;
        }
        if (lane == 63) lds4[w] = incl;
        __syncthreads();
        uint32_t before = carry_s;
        for (uint32_t k = 0; k < w; k++) before += lds4[k];
        if (i < nblocks) counts[1u + i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == kCompactThreads - 1u) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0].index = carry_s;
        out[0].val = head_val;
        counts[0] = 0u;                     // ticket ready for the next run
        if (reset_word) *reset_word = 0u;   // e.g. the SpMSpV chunk-queue counter, ready for the next run
        dir.decide(carry_s);
    }
}

// `own_offsets`: long lists.  offsets[1 ..] still hold the blocks' raw counts: every block sums the counts in front of it
// itself (a few loads per thread) instead of waiting for a one-block scan launch between the two passes, and the last
// block writes the head element and takes the loop decision -- one dependent launch less per SpMSpV (~4 us).
template <typename Src>
__global__ __launch_bounds__(256) void compact_write_kernel(Src src, const uint32_t *__restrict__ offsets,
                                                            gl_idx_val *__restrict__ out, Gate gate, bool own_offsets, float head_val,
                                                            uint32_t *__restrict__ reset_word, Direction dir) {
    __shared__ uint32_t lds4[4];
    if (gate.closed()) return;
    const uint32_t n = src.size();
    const uint32_t base = blockIdx.x * kCompactChunk;
    uint32_t pos;
    if (own_offsets) {
        uint32_t part = 0;
        for (uint32_t i = threadIdx.x; i < blockIdx.x; i += kCompactThreads) part += offsets[1u + i];
        for (int dlt = 32; dlt >= 1; dlt >>= 1) part += __shfl_down(part, dlt);
        if ((threadIdx.x & 63u) == 0) lds4[threadIdx.x >> 6] = part;
        __syncthreads();
        pos = lds4[0] + lds4[1] + lds4[2] + lds4[3];
        __syncthreads();
        if (blockIdx.x == gridDim.x - 1u && threadIdx.x == 0) {   // total = everything in front of the last block + its own count
            const uint32_t total = pos + offsets[1u + blockIdx.x];
            out[0].index = total;
            out[0].val = head_val;
            if (reset_word) *reset_word = 0u;
            dir.decide(total);
        }
        if (base >= n) return;
    } else {
        if (base >= n) return;
        pos = offsets[1u + blockIdx.x];
    }
#pragma unroll
    for (uint32_t j = 0; j < kCompactItems; j++) {
        uint32_t i = base + j * kCompactThreads + threadIdx.x;
        gl_idx_val item;
        bool keep = false;
        if (i < n) {
            keep = src.get(i, item);
            src.consumed(i);
        }
        uint32_t total;
        uint32_t rank = block_rank_256(keep, lds4, &total);
        if (keep) {
            out[1u + pos + rank] = item;
            src.emitted(item);
        }
        pos += total;
    }
}

// Runs the passes for at most `max_items` candidates (host-side bound for the grid).
template <typename Src>
static int run_compaction(Src src, uint32_t max_items, uint32_t *d_counts, gl_idx_val *d_out, float head_val,
                          hipStream_t s, uint32_t *d_reset_word = nullptr, Gate gate = Gate(), Direction dir = Direction()) {
    uint32_t nblocks = cdiv(max_items, kCompactChunk);
    if (nblocks == 0) nblocks = 1;
    const bool fused = nblocks <= kCompactFuseBlocks;
    compact_count_kernel<Src><<<nblocks, kCompactThreads, 0, s>>>(src, d_counts, d_out, head_val, d_reset_word, fused, gate, dir);
    GL_LAUNCH_CHECK();
    compact_write_kernel<Src><<<nblocks, kCompactThreads, 0, s>>>(src, d_counts, d_out, gate, !fused, head_val, d_reset_word, dir);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// grow-only device scratch owned by the runtime (used where no plan exists to hold workspace)
int scratch_reserve(size_t bytes, void **d_ptr);

}  // namespace gl

#endif  // GL_COMPACT_H_
