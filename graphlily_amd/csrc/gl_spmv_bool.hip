// SpMV for the (||,&&) semiring only (GL_PLAN_BOOLEAN): y[r] = mask( zero || OR_c (A[r,c] && x[c]) ).
//
// This is the BFS pull step (app/bfs.h:107-128 -> SpMVModule::run with LogicalSemiring).  With boolean
// algebra neither the matrix values nor the float vector have to travel:
//   entries   4 bytes: x word offset from the group's base : 13 | row_in_block : 14 | bit in the word : 5,
//             column-sorted per row block like the general
//             layout (gl_spmv.hip); entries whose value is 0 are dropped at format time (a && b is false).
//             A group is 256 entries = 1 KB, one coalesced 16-byte-per-lane read; half the HBM bytes.
//   x         packed to one bit per column by spmv_bool_pack_kernel (12 MB read once per run), and a whole
//             "phase" of 1 179 648 columns (144 KB) is LDS-resident while a workgroup sweeps it, so there is
//             no vector-memory gather at all: the lookups are LDS reads.
//   tile      one bit per row of the block (ds_or_b32).
// What is left is the 4-byte stream at HBM speed.  Units, segment-major numbering, direct epilogue for
// unsplit blocks and the init kernel + plain stores (1.0f, idempotent) for split ones are as in gl_spmv.hip.
//
// Round 6: the stream is DELTA-CODED to 3 bytes per entry when every group allows it (bool_plan_compress, a pass over the
// formatted groups on the device, whichever formatter wrote them): a group's 256 entries are sorted by column, so an entry keeps
// its 16-bit row slot and the 8-bit distance to its predecessor's bit index -- lane l holds entries 4 l .. 4 l + 3: 8 bytes of
// slots + 4 bytes of deltas, one global_load_dwordx3 per lane and group, 768 bytes instead of 1024 -- and the kernel rebuilds the
// indices with one v_sad_u8 (a lane's four deltas), ONE 32-bit prefix scan over the lanes (six DPP adds per 256 entries) and three
// adds.  Padding entries repeat their predecessor's index with the ghost row slot.  Deltas of 256..1023 put bits 8..9 into the
// two spare bits of the row slot (a second decoder, three more instructions per entry: pokec's rare columns); a plan with a gap
// of more than 1023 columns inside some group keeps the 4-byte form.
#include "gl_spmv_plan.h"
#include "gl_bfs_shard.h"

namespace gl {

struct BoolArgs {
    const void *entries;       // groups of kBoolGroup: lane l holds entries kBoolLane * l ... (one 8- or 16-byte load)
    const uint32_t *bases;     // per group: first x word (of its phase) the group's word offsets count from
    const uint4 *units;        // 2 per unit: {first span, #spans, first row, #rows | direct << 31}, {hub offset, #hub rows, -, -}
    const uint32_t *hub_rows;  // row_in_block of every hub row, per block
    const uint4 *spans;        // {first xbits word of the phase, first group, end group, lo4 | hi4 << 16}
    const uint32_t *xbits;
    const float *mask;
    float *y;
    float zero;
    const uint32_t *run_flag;  // non-null: the launch is a no-op unless run_flag[0] != 0 (gl_spmspv_run's direction switch)
    // fused BFS pull step (gl_bfs_pull_step): rows reached now and not visited before get `level` and form the next
    // frontier, written as bits
    uint32_t *bits_out;
    float *dist;
    float level;
    uint32_t tickets;          // 1: ring slots are refilled from an LDS ticket per span; 0: static split (A/B builds)
    const uint32_t *gate = nullptr;   // non-null: the launch is a no-op unless gate[0] (gate_op) gate_value (gl_bfs_pull_step_gated)
    uint32_t gate_value = 0;
    int gate_op = GL_GATE_EQ;
    // pull -> push decision of a device-resident BFS schedule (gl_bfs_pull_step_back): the fused epilogue counts the rows
    // it puts into the next frontier (ctl[5]); the workgroup that finishes last (ticket ctl[6]) compares the count with
    // back_threshold * n and, if the frontier has become that small and iterations remain, re-opens the push gate
    // (ctl[0] = 0xffffffff) and marks the slot (ctl[4]) so that the list-building pass behind this launch runs
    uint32_t *back_ctl = nullptr;
    uint32_t back_slot = 0, back_may_continue = 0, back_n = 1;
    float back_threshold = 0.0f;
    // bit-frontier BFS schedule (gl_bfs_bits_pull_step, BfsBitsCtl in gl_common.h): the launch runs when its slot pulls
    // OR when the slot's push goes row-wise -- the same pass either way; the fused epilogue also sums the column lengths
    // of the rows it adds (the next push's work) and the workgroup that finishes last takes the step's decisions
    BfsBitsCtl v2;
    const uint32_t *v2_indptr = nullptr;   // column pointers of the CSC plan of the same matrix
    uint32_t v2_ncols = 0;
    uint32_t *v2_push_acc = nullptr;       // totals of the slot's push step (64 lines of 32 words), when that one ran
    const uint32_t *v2_rowptr = nullptr;   // row pointers of this plan's own CSR copy (row lengths: the bottom-up bookkeeping)
    // row-sharded schedule: the launch runs or not by the same control words, but keeps no totals and decides nothing
    // (gl_bfs_bits_decide does, on the all-gathered frontier)
    bool v2_deferred = false;
    // one-launch shard step (gl_bfs_shard.h): the epilogue tallies what it adds to the next frontier -- vertices, their
    // GLOBAL column lengths (t_col_len) and their row lengths (v2_rowptr, indexed by row - v2_row_base) -- into the rank's lines
    uint32_t *tally_mine = nullptr;
    const uint32_t *t_col_len = nullptr;
    uint32_t v2_row_base = 0;
};

// x != 0 packed little-endian, 64 columns per wavefront step; words past num_cols are zero
__global__ __launch_bounds__(256) void spmv_bool_pack_kernel(const float *__restrict__ x, uint32_t num_cols,
                                                             uint64_t *__restrict__ bits, uint32_t nwords64) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nwaves = gridDim.x * 4u;
    for (uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6); w < nwords64; w += nwaves) {
        const uint32_t c = w * 64u + lane;
        const bool set = (c < num_cols) && (x[c] != 0.0f);
        const uint64_t m = __ballot(set);
        if (lane == 0) bits[w] = m;
    }
}

// the inverse: x[c] = bit c ? 1.0f : 0.0f (a frontier kept as bits handed back to a caller that reads the float vector)
__global__ __launch_bounds__(256) void spmv_bool_unpack_kernel(const uint32_t *__restrict__ bits, uint32_t n, float *__restrict__ x) {
    for (uint32_t c = blockIdx.x * 256u + threadIdx.x; c < n; c += gridDim.x * 256u) x[c] = ((bits[c >> 5] >> (c & 31u)) & 1u) ? 1.0f : 0.0f;
}

// gl_bfs_bits_begin_from: control words of a schedule that pulls in every slot + three rotating bit vectors, the first packed
// from the caller's float frontier (the caller's distances are left alone)
__global__ __launch_bounds__(256) void bfs_bits_begin_from_kernel(uint32_t *__restrict__ ctl, uint32_t ctl_words, const float *__restrict__ x,
                                                                  uint32_t n, uint64_t *__restrict__ bits, uint32_t words64) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if (tid < ctl_words) ctl[tid] = tid == 4u ? 0xffffffffu : (tid == 15u ? ctl_words : 0u);   // [0] = 0: every slot pulls
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6); w < 3u * words64; w += gridDim.x * 4u) {
        uint64_t m = 0ull;
        if (w < words64) {
            const uint32_t c = w * 64u + lane;
            m = __ballot(c < n && x[c] != 0.0f);
        }
        if (lane == 0) bits[w] = m;
    }
}

// ... and, for a schedule that starts in the middle of a BFS (the reference's pull_push hands over to pulling after a few push
// iterations, app/bfs.h:195-216): the non-zeros in the rows reached so far, which the steps' bottom-up decision goes by
// (BfsBitsCtl: ctl[12..13]).  Runs behind bfs_bits_begin_from_kernel.
__global__ __launch_bounds__(256) void bfs_bits_visited_kernel(uint32_t *__restrict__ ctl, const float *__restrict__ distance, uint32_t n,
                                                               const uint32_t *__restrict__ row_ptr) {
    unsigned long long sum = 0ull;
    for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < n; r += gridDim.x * 256u)
        if (distance[r] != 0.0f) sum += row_ptr[r + 1u] - row_ptr[r];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d);
    if ((threadIdx.x & 63u) == 0u && sum) atomicAdd(reinterpret_cast<unsigned long long *>(ctl + 12), sum);
}

#ifndef GL_BOOL_STEP
#define GL_BOOL_STEP 2
#endif
constexpr int kBoolStep = GL_BOOL_STEP;   // ring slots processed together (divides the ring depth)
constexpr int kBoolLane = kBoolGroup / 64;   // entries per lane and load
struct BoolElem {   // one lane's share of a group
    uint32_t v[kBoolLane];
};
constexpr uint32_t kBoolGroupBytesC = kBoolGroup * 3u;   // compressed group: 2 B of row slot + 1 B of delta per entry
// KEEP bit 0: no non-temporal hint -- plans that fit the Infinity Cache stay there between runs (gl_common.h);
// KEEP bit 1: the delta-coded 3-byte stream (v[0], v[1]: four 16-bit row slots, v[2]: four 8-bit deltas, v[3] unused; 12 bytes per lane);
// KEEP bit 2 (with bit 1): some delta of the plan needs 10 bits, and bits 8..9 sit on top of the 14-bit row slots
template <int KEEP>
__device__ __forceinline__ BoolElem bool_load(const void *entries, size_t group, uint32_t lane) {
    BoolElem e;
    if ((KEEP & 2) && kBoolLane == 4) {
        // a lane's 12 bytes are contiguous: ONE global_load_dwordx3 per lane and group (4-byte aligned)
        typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
        typedef u32x3_t __attribute__((aligned(4))) u32x3_packed_t;
        const u32x3_packed_t *q =
            reinterpret_cast<const u32x3_packed_t *>(static_cast<const unsigned char *>(entries) + group * kBoolGroupBytesC + lane * 12u);
        const u32x3_t t = (KEEP & 1) ? *q : __builtin_nontemporal_load(q);
        e.v[0] = t.x, e.v[1] = t.y, e.v[2] = t.z;
        e.v[kBoolLane - 1] = 0u;
        return e;
    }
    if (kBoolLane == 4) {
        const uint4 *q = static_cast<const uint4 *>(entries) + group * 64u + lane;
        const uint4 t = (KEEP & 1) ? load_stream_keep16(q) : load_stream_nt16(q);
        e.v[0] = t.x, e.v[1] = t.y, e.v[kBoolLane - 2] = t.z, e.v[kBoolLane - 1] = t.w;
    } else {
        const uint2 *q = static_cast<const uint2 *>(entries) + group * 64u + lane;
        const uint2 t = (KEEP & 1) ? load_stream_keep(q) : load_stream_nt(q);
        e.v[0] = t.x, e.v[1] = t.y;
    }
    return e;
}

// inclusive prefix sum of v over the 64 lanes: four row_shr steps scan the 16-lane rows, row_bcast:15 / :31 carry the row totals on
__device__ __forceinline__ uint32_t bool_wave_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2 and 3
    return v;
}

// the kernel's body for workgroup `unit` (also run by the one-launch shard step of the BFS schedule, below)
template <int MASK, int U, int FUSED, int KEEP>
__device__ __forceinline__ void spmv_bool_body(const BoolArgs &a, uint32_t *lds_words, const uint32_t unit) {
    // tile first: its byte offsets fit the 16-bit immediate of the LDS instructions either way
    uint32_t *tile = lds_words;                     // kBoolTileWords: one bit per row slot, slot 16383 = padding
    uint32_t *xw = lds_words + kBoolTileWords;      // kBoolPhaseWords

    if (a.run_flag && load_const(a.run_flag) == 0u) return;
    if (a.v2.ctl && (a.v2.finished() || a.v2.scatters() || a.v2.bottom_up())) {
        // the slot's push step ran (it is enqueued in front of this launch) -- scattering, or as the bottom-up pull: add up
        // its totals and take its decisions.  decide() does not change what scatters() / bottom_up() say about THIS slot,
        // so the other workgroups may look later.
        if (!a.v2_deferred && unit == 0 && threadIdx.x < 64u) {
            uint32_t *line = a.v2_push_acc + 32u * threadIdx.x;
            uint32_t fresh = line[0];
            unsigned long long work = *reinterpret_cast<unsigned long long *>(line + 2);
            unsigned long long work_rows = *reinterpret_cast<unsigned long long *>(line + 6);
            if (fresh) {
                line[0] = 0u;
                *reinterpret_cast<unsigned long long *>(line + 2) = 0ull;
                *reinterpret_cast<unsigned long long *>(line + 6) = 0ull;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                fresh += __shfl_down(fresh, d);
                work += __shfl_down(work, d);
                work_rows += __shfl_down(work_rows, d);
            }
            if (threadIdx.x == 0) a.v2.decide(fresh, work, work_rows);
        }
        return;
    }
    if (a.gate) {   // (a plain load: the word is written by kernels earlier in the stream)
        const uint32_t w = *a.gate;
        if (!(a.gate_op == GL_GATE_EQ ? w == a.gate_value : a.gate_op == GL_GATE_GT ? w > a.gate_value : w <= a.gate_value)) return;
    }
    if (a.v2.ctl && unit == 0 && threadIdx.x == 0) a.v2.record_mode(2u);   // this slot streams the matrix row-wise
    const uint4 d = a.units[2u * unit], dh = a.units[2u * unit + 1u];
    const uint32_t span0 = d.x, nspans = d.y, row0 = d.z;
    const uint32_t nrows = d.w & 0xffffu;
    const bool direct = (d.w >> 31) != 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    for (uint32_t i = threadIdx.x; i < kBoolTileWords; i += kThreads) tile[i] = 0u;
    // one ticket counter per span (a unit has at most kBoolMaxPhases of them): the first kWaves * U groups of a span are
    // taken by wave and ring slot, the rest is drawn kBoolStep groups at a time, so that the wavefronts of a workgroup
    // run out of work together (a static split lets the hardware's oldest-first issue finish the low wavefronts early)
    __shared__ uint32_t next_group[kBoolMaxPhases];
    if (threadIdx.x < kBoolMaxPhases) next_group[threadIdx.x] = kWaves * U;

    for (uint32_t sp = 0; sp < nspans; sp++) {
        const uint4 s = load_const(a.spans + span0 + sp);
        // Register ring: U groups per wavefront are always in flight.  A slot is refilled right after it has
        // been consumed, every load is unconditional (indices clamp to the span's last group) so that the
        // compiler's s_waitcnt vmcnt(N) stay exact -- a batch "load U, wait, process U" loop leaves the
        // memory pipe empty for the whole processing phase and is latency-bound at ~2/3 of HBM speed.
        // The ring is primed BEFORE the x bits of the phase are copied: the copy hides the HBM latency.
        const uint32_t ngroups = s.z - s.y, glast = s.z - 1u;
        BoolElem e[U];
        uint32_t b[U], li[U];   // li: the slot's group, counted from the span's first
#pragma unroll
        for (int u = 0; u < U; u++) {
            li[u] = wave + u * kWaves;
            const uint32_t gi = min(s.y + li[u], glast);
            e[u] = bool_load<KEEP>(a.entries, gi, lane);
            b[u] = load_const(a.bases + gi);
            // keep slot order = issue order: if the scheduler reverses these loads, slot 0 becomes the youngest
            // and the loop header needs vmcnt(0), which empties the ring once per iteration
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();   // the previous phase's lookups are done (first pass: tile is zeroed)
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.xbits + s.x);
            uint4 *dst = reinterpret_cast<uint4 *>(xw);
            for (uint32_t i = (s.w & 0xffffu) + threadIdx.x; i < (s.w >> 16); i += kThreads) dst[i] = src[i];
        }
        __syncthreads();
        uint32_t drawn = 0;   // lane 0: the ticket for the next refill, drawn one step ahead
        if (a.tickets && lane == 0) drawn = __hip_atomic_fetch_add(&next_group[sp], (uint32_t)kBoolStep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (li[0] < ngroups) {   // slot 0 holds the oldest group; younger slots past the end re-apply the last group
#pragma unroll
            for (int u = 0; u < U; u += kBoolStep) {
                // Per entry: bit index = base + offset, one LDS word read, bit test; hits OR the row's bit into the
                // tile.  Padding entries are ordinary entries of the ghost row slot 16383, which the epilogue never
                // reads, and a slot past the span's end holds the span's last group again (clamped index), which
                // is harmless to apply twice -- so there is no validity test at all.  16-lane SIMDs make every VALU
                // instruction cost 4 clocks per wavefront: the path is kept to ~6 of them per entry, and kBoolStep ring
                // slots are processed together so that kBoolLane * kBoolStep LDS lookups are in flight before the first test.
                // entry = word offset : 13 | row slot : 14 | bit : 5 (v_bfe takes the bit number from the low 5 bits
                // of its operand, so the entry itself is the bit selector); b = first x word of the group
                constexpr int NE = kBoolLane * kBoolStep;
                uint32_t v[NE], w[NE];
                if constexpr ((KEEP & 2) != 0 && kBoolLane == 4) {
                    // delta-coded group: v[k] becomes the entry's bit index relative to the group's first x word -- the lane's
                    // four deltas summed (v_sad_u8), ONE prefix scan over the lanes, three running adds -- and rr[k] its row slot
                    uint32_t rr[NE];
#pragma unroll
                    for (int g = 0; g < kBoolStep; g++) {
                        const BoolElem &el = e[u + g];
                        const uint32_t dl = el.v[2];
                        if constexpr ((KEEP & 4) != 0) {
                            // 10-bit deltas: bits 8..9 ride in the two spare bits of the 16-bit row slots -- the slots' high bytes
                            // gathered into one word (v_perm), their top two bits each
                            const uint32_t hh = (__builtin_amdgcn_perm(el.v[1], el.v[0], 0x07050301u) >> 6) & 0x03030303u;
                            const uint32_t sum = __builtin_amdgcn_sad_u8(dl, 0u, __builtin_amdgcn_sad_u8(hh, 0u, 0u) << 8);
                            const uint32_t before = bool_wave_scan(sum) - sum;
                            v[4 * g] = before + (dl & 255u) + ((hh & 3u) << 8);
                            v[4 * g + 1] = v[4 * g] + ((dl >> 8) & 255u) + (((hh >> 8) & 3u) << 8);
                            v[4 * g + 2] = v[4 * g + 1] + ((dl >> 16) & 255u) + (((hh >> 16) & 3u) << 8);
                            v[4 * g + 3] = v[4 * g + 2] + (dl >> 24) + ((hh >> 24) << 8);
                            rr[4 * g] = el.v[0] & kRowPad, rr[4 * g + 1] = (el.v[0] >> 16) & kRowPad;
                            rr[4 * g + 2] = el.v[1] & kRowPad, rr[4 * g + 3] = (el.v[1] >> 16) & kRowPad;
                        } else {
                            const uint32_t sum = __builtin_amdgcn_sad_u8(dl, 0u, 0u);
                            const uint32_t before = bool_wave_scan(sum) - sum;
                            v[4 * g] = before + (dl & 255u);
                            v[4 * g + 1] = v[4 * g] + ((dl >> 8) & 255u);
                            v[4 * g + 2] = v[4 * g + 1] + ((dl >> 16) & 255u);
                            v[4 * g + 3] = v[4 * g + 2] + (dl >> 24);
                            rr[4 * g] = el.v[0] & 0xffffu, rr[4 * g + 1] = el.v[0] >> 16;
                            rr[4 * g + 2] = el.v[1] & 0xffffu, rr[4 * g + 3] = el.v[1] >> 16;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NE; k++) w[k] = xw[b[u + k / kBoolLane] + (v[k] >> 5)];
#pragma unroll
                    for (int k = 0; k < NE; k++)
                        if (__builtin_amdgcn_ubfe(w[k], v[k], 1u)) atomicOr(&tile[rr[k] >> 5], 1u << (rr[k] & 31u));
                } else {
#pragma unroll
                for (int k = 0; k < NE; k++) v[k] = e[u + k / kBoolLane].v[k % kBoolLane];
#pragma unroll
                for (int k = 0; k < NE; k++) w[k] = xw[b[u + k / kBoolLane] + (v[k] >> 19)];
#pragma unroll
                for (int k = 0; k < NE; k++)
                    if (__builtin_amdgcn_ubfe(w[k], v[k], 1u)) {
                        const uint32_t r = (v[k] >> 5) & kRowPad;
                        atomicOr(&tile[r >> 5], 1u << (r & 31u));
                    }
                }
                const uint32_t t = __builtin_amdgcn_readfirstlane(drawn);
                if (a.tickets && lane == 0) drawn = __hip_atomic_fetch_add(&next_group[sp], (uint32_t)kBoolStep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (int k = 0; k < kBoolStep; k++) {
                    li[u + k] = a.tickets ? t + k : li[u + k] + kWaves * U;
                    const uint32_t gn = min(s.y + li[u + k], glast);
                    e[u + k] = bool_load<KEEP>(a.entries, gn, lane);
                    b[u + k] = load_const(a.bases + gn);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < kBoolLane; k++) asm volatile("" : : "v"(e[u].v[k]));   // retire the clamped tail loads
    }
    __syncthreads();

    if (dh.y) {   // hub rows: OR their 32 private bits (one per tile word 480..511) back into the row's bit
        if (threadIdx.x < dh.y) {
            uint32_t any = 0u;
#pragma unroll
            for (uint32_t j = 0; j < kBoolHubSlots; j++) any |= tile[kBoolHubBit0 / 32u + j] >> threadIdx.x;
            if (any & 1u) {
                const uint32_t r = a.hub_rows[dh.x + threadIdx.x];
                atomicOr(&tile[r >> 5], 1u << (r & 31u));
            }
        }
        __syncthreads();
    }

    if (FUSED == 2) {
        // The same step for SPLIT plans (several units share a block's rows, each with a part of the columns) and only inside
        // the bit-frontier schedule, whose next-frontier vector starts all zero: a row this unit reached whose distance is
        // still 0 sets its bit, and the unit whose atomicOr set it first writes the level -- as the push step does.  (A unit
        // that reads a distance another unit has just written sees `level`, not 0, and leaves the row alone: same result.)
        // 64 rows per wavefront step (blocks start on multiples of 64 rows): one coalesced distance load, ONE atomicOr per
        // 32-bit word of the output vector with the wavefront's combined mask -- a dense frontier would otherwise cost an
        // atomic per row and unit (orkut's heavy slot on a 1/8 shard: 3 M of them)
        uint32_t t_fresh = 0u, t_work = 0u, t_rows = 0u;   // one-launch shard step: this lane's claims (<= 16 rows)
        // Four row groups per step, every load / atomic of a stage issued before the first use of its result: a group is a
        // chain of dependent round trips (distance, atomicOr, then -- tallies -- the row's lengths), and a 1/8 shard's units
        // share 12 K rows each: one group at a time the epilogue took as long as the stream
        constexpr int E2 = 4;
        for (uint32_t i0 = (threadIdx.x >> 6) * 64u; i0 < nrows; i0 += E2 * kThreads) {
            bool cand[E2];
            float dv[E2];
#pragma unroll
            for (int u = 0; u < E2; u++) {
                const uint32_t i = i0 + u * kThreads + lane;
                cand[u] = i < nrows && ((tile[i >> 5] >> (i & 31u)) & 1u);
                dv[u] = cand[u] ? a.dist[row0 + i] : 1.0f;
            }
            uint32_t old[E2];
#pragma unroll
            for (int u = 0; u < E2; u++) {
                cand[u] = cand[u] && dv[u] == 0.0f;
                const uint64_t m = __ballot(cand[u]);
                const uint32_t half = lane >> 5, mine = (uint32_t)(m >> (32u * half));
                old[u] = 0u;
                if ((lane & 31u) == 0u && mine) old[u] = atomicOr(&a.bits_out[(row0 + i0 + u * kThreads) / 32u + half], mine);
            }
            uint32_t cl[E2], r0[E2], r1[E2];
#pragma unroll
            for (int u = 0; u < E2; u++) {
                const uint32_t row = row0 + i0 + u * kThreads + lane;
                const uint32_t o = __shfl(old[u], (int)((lane >> 5) * 32u));
                cand[u] = cand[u] && !((o >> (lane & 31u)) & 1u);     // this unit set the row's bit first
                const bool t = cand[u] && a.tally_mine != nullptr;
                cl[u] = t ? a.t_col_len[row] : 0u;
                r0[u] = (t && a.v2_rowptr) ? a.v2_rowptr[row - a.v2_row_base] : 0u;
                r1[u] = (t && a.v2_rowptr) ? a.v2_rowptr[row - a.v2_row_base + 1u] : 0u;
            }
#pragma unroll
            for (int u = 0; u < E2; u++) {
                if (cand[u]) {
                    a.dist[row0 + i0 + u * kThreads + lane] = a.level;
                    t_fresh += 1u;
                    t_work += cl[u];
                    t_rows += r1[u] - r0[u];
                }
            }
        }
        if (a.tally_mine) tally_block_add(a.tally_mine, unit, t_fresh, t_work, t_rows);
    } else if (FUSED) {
        // SpMV masked by `distance == 0`, eWiseAdd(+0), assign(level) where the result is set, and the packing of
        // the next frontier (app/bfs.h:118-123) in one epilogue: 64 rows per wavefront step, one 64-bit word out.
        // Blocks of boolean plans start on multiples of 64 rows, so every word has exactly one writer.
        uint32_t nfresh = 0;   // lane 0 of every wavefront: rows this wavefront put into the next frontier
        uint32_t work = 0u;    // v2: column lengths of this lane's fresh rows (a lane sees <= 15 rows: no overflow below 2^28 each)
        uint32_t work_rows = 0u;   // ... and their row lengths
        // Four row groups per step, every load of a stage issued before the first use: a row is three dependent round trips
        // (distance, then -- bit-frontier schedule only -- its two column pointers), and one at a time they made the epilogue
        // a sixth of the launch.  No two threads touch the same row, so the stores of a step cannot feed its loads.
        constexpr int E = 4;
        for (uint32_t i0 = (threadIdx.x >> 6) * 64u; i0 < nrows; i0 += E * kThreads) {
            bool hit[E];
            float dv[E];
#pragma unroll
            for (int u = 0; u < E; u++) {
                const uint32_t i = i0 + u * kThreads + lane;
                hit[u] = i < nrows && ((tile[i >> 5] >> (i & 31u)) & 1u);
                dv[u] = hit[u] ? a.dist[row0 + i] : 1.0f;
            }
            uint32_t p0[E], p1[E], r0[E], r1[E];
#pragma unroll
            for (int u = 0; u < E; u++) {
                const uint32_t row = row0 + i0 + u * kThreads + lane;
                hit[u] = hit[u] && dv[u] == 0.0f;       // fresh: reached now, never before
                const bool want = hit[u] && a.v2_indptr != nullptr && row < a.v2_ncols;
                p0[u] = want ? a.v2_indptr[row] : 0u;
                p1[u] = want ? a.v2_indptr[row + 1u] : ((hit[u] && a.t_col_len) ? a.t_col_len[row] : 0u);
                const bool wantr = hit[u] && a.v2_rowptr != nullptr;
                r0[u] = wantr ? a.v2_rowptr[row - a.v2_row_base] : 0u;
                r1[u] = wantr ? a.v2_rowptr[row - a.v2_row_base + 1u] : 0u;
            }
#pragma unroll
            for (int u = 0; u < E; u++) {
                const uint32_t g0 = i0 + u * kThreads;   // wave-uniform
                if (g0 < nrows) {
                    if (hit[u]) a.dist[row0 + g0 + lane] = a.level;
                    const uint64_t m = __ballot(hit[u]);
                    if (lane == 0) reinterpret_cast<uint64_t *>(a.bits_out)[(row0 + g0) >> 6] = m;
                    nfresh += (uint32_t)__popcll(m);
                    work += p1[u] - p0[u];
                    work_rows += r1[u] - r0[u];
                }
            }
        }
        if (a.tally_mine) {
            // (nfresh is lane 0's count of the wavefront's rows; work / work_rows are per lane)
            tally_block_add(a.tally_mine, unit, lane == 0 ? nfresh : 0u, work, work_rows);
        } else if (a.v2.ctl && !a.v2_deferred) {
            __shared__ uint32_t v2_fresh_s, v2_last_s;
            __shared__ unsigned long long v2_work_s, v2_rows_s;
            if (threadIdx.x == 0) {
                v2_fresh_s = 0u;
                v2_work_s = 0ull;
                v2_rows_s = 0ull;
            }
            __syncthreads();
            unsigned long long work64 = work, rows64 = work_rows;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                work64 += __shfl_down(work64, d);
                rows64 += __shfl_down(rows64, d);
            }
            if (lane == 0 && nfresh) {
                atomicAdd(&v2_fresh_s, nfresh);
                atomicAdd(&v2_work_s, work64);
                atomicAdd(&v2_rows_s, rows64);
            }
            __syncthreads();
            // Totals and "who is last" through the 64 accumulator lines of the push step (idle in a slot whose pull step
            // runs), then one root ticket per line: all 256 workgroups end within microseconds of each other, and three
            // atomics each on ONE word cost ~5 us per launch (measured: BFS.pull on 23 slots +110 us).
            if (threadIdx.x == 0) {
                const uint32_t l = unit & 63u, nlines = min(gridDim.x, 64u);
                uint32_t *line = a.v2_push_acc + 32u * l;
                if (v2_fresh_s) {
                    atomicAdd(line, v2_fresh_s);
                    atomicAdd(reinterpret_cast<unsigned long long *>(line + 2), v2_work_s);
                    if (v2_rows_s) atomicAdd(reinterpret_cast<unsigned long long *>(line + 6), v2_rows_s);
                }
                __threadfence();
                bool last = atomicAdd(line + 4, 1u) == (gridDim.x - l + 63u) / 64u - 1u;   // last workgroup of this line
                if (last) {
                    __threadfence();
                    last = atomicAdd(&a.v2.ctl[6], 1u) == nlines - 1u;                      // ... of the launch
                }
                v2_last_s = last ? 1u : 0u;
            }
            __syncthreads();
            if (v2_last_s && threadIdx.x < 64u) {   // one wavefront adds up the lines (one thread doing it: +8 us per launch)
                __threadfence();
                const uint32_t nlines = min(gridDim.x, 64u);
                uint32_t total = 0u;
                unsigned long long wk = 0ull, wr = 0ull;
                if (threadIdx.x < nlines) {
                    uint32_t *ln = a.v2_push_acc + 32u * threadIdx.x;
                    total = __hip_atomic_load(ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    wk = __hip_atomic_load(reinterpret_cast<unsigned long long *>(ln + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    wr = __hip_atomic_load(reinterpret_cast<unsigned long long *>(ln + 6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ln[0] = 0u;
                    *reinterpret_cast<unsigned long long *>(ln + 2) = 0ull;
                    *reinterpret_cast<unsigned long long *>(ln + 6) = 0ull;
                    ln[4] = 0u;
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    total += __shfl_down(total, d);
                    wk += __shfl_down(wk, d);
                    wr += __shfl_down(wr, d);
                }
                if (threadIdx.x == 0) {
                    a.v2.ctl[6] = 0u;
                    a.v2.decide(total, wk, wr);
                }
            }
        } else if (a.back_ctl) {
            __shared__ uint32_t fresh_s;   // one global atomic per workgroup, not per wavefront
            if (threadIdx.x == 0) fresh_s = 0u;
            __syncthreads();
            if (lane == 0 && nfresh) atomicAdd(&fresh_s, nfresh);
            __syncthreads();
            if (threadIdx.x == 0) {
                if (fresh_s) atomicAdd(&a.back_ctl[5], fresh_s);
                __threadfence();
            }
            if (threadIdx.x == 0 && atomicAdd(&a.back_ctl[6], 1u) == gridDim.x - 1u) {   // the last workgroup decides
                __threadfence();
                const uint32_t total = __hip_atomic_load(&a.back_ctl[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.back_ctl[5] = 0u;
                a.back_ctl[6] = 0u;
                if (a.back_may_continue != 0u && (float)total / (float)a.back_n < a.back_threshold) {
                    a.back_ctl[0] = 0xffffffffu;     // pushing again from the next slot on
                    a.back_ctl[4] = a.back_slot;     // ... and this slot's new frontier is wanted as a list
                    a.back_ctl[7] = __float_as_uint(a.back_threshold);   // Direction::decide: stay while below this
                }
            }
        }
    } else if (direct) {
        for (uint32_t i = threadIdx.x; i < nrows; i += kThreads) {
            const uint32_t row = row0 + i;
            const bool hit = (tile[i >> 5] >> (i & 31u)) & 1u;
            float out = (a.zero != 0.0f || hit) ? 1.0f : 0.0f;
            if (MASK != GL_NOMASK) {
                if (!mask_allows<MASK>(a.mask[row], 0.0f)) out = 0.0f;   // spmv_module.h:518-530
            }
            a.y[row] = out;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < nrows; i += kThreads) {
            if (!((tile[i >> 5] >> (i & 31u)) & 1u)) continue;
            const uint32_t row = row0 + i;
            if (MASK != GL_NOMASK) {
                if (!mask_allows<MASK>(a.mask[row], 0.0f)) continue;
            }
            a.y[row] = 1.0f;   // every writer stores the same value
        }
    }
}

static uint32_t bool_tickets() {
    static const uint32_t t = 1u;
    return t;
}

#ifndef GL_BOOL_U
#define GL_BOOL_U 4
#endif
// ring depth.  Same-box sweep (orkut / products, masked, x density 0.5 and 0.02): groups of 256 entries (16-byte loads),
// two slots per step, four slots deep 0.160 / 0.097 ms; 128-entry groups (8-byte loads), 2, 6: 0.170 / 0.102 ms.
constexpr int kBoolUnroll = GL_BOOL_U;
template <int MASK, int U, int FUSED, int KEEP = 0>
__global__ __launch_bounds__(kThreads) void spmv_bool_kernel(BoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_words[];
    spmv_bool_body<MASK, U, FUSED, KEEP>(a, lds_words, blockIdx.x);
}

// One slot of the bit-frontier BFS schedule on a row shard in ONE launch (gl_bfs_shard.h): the previous slot's decision
// from all ranks' tallies, then the slot's step -- scattering push, bottom-up scan or streaming pull.
template <int U, int FUSED, int KEEP>
__global__ __launch_bounds__(kThreads) void bfs_shard_step_kernel(BoolArgs ba, BfsPushArgs pa, BfsShardArgs sa) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_words[];
    __shared__ uint32_t s_state[16];
    const BfsBitsCtl c = shard_prologue(sa, s_state);
    if (sa.finish || c.finished()) return;
    const bool first = blockIdx.x == 0 && threadIdx.x == 0;
    if (c.scatters()) {
        if (first) c.record_mode(1u);
        if (blockIdx.x >= sa.push_blocks) return;
        uint32_t fresh = 0u, work = 0u, work_rows = 0u;
        bfs_shard_scatter<kThreads>(pa, lds_words, blockIdx.x, sa.push_blocks, fresh, work, work_rows);
        tally_block_add(sa.tally_mine, blockIdx.x, fresh, work, work_rows);
        return;
    }
    if (pa.row_idx && c.bottom_up()) {
        if (first) c.record_mode(3u);
        uint32_t fresh = 0u, work = 0u, work_rows = 0u;
        bfs_shard_bottom_up(pa, blockIdx.x * kWaves + (threadIdx.x >> 6), gridDim.x * kWaves, fresh, work, work_rows);
        tally_block_add(sa.tally_mine, blockIdx.x, fresh, work, work_rows);
        return;
    }
    if (first) c.record_mode(2u);
    if (blockIdx.x < sa.pull_units) spmv_bool_body<GL_NOMASK, U, FUSED, KEEP>(ba, lds_words, blockIdx.x);
}

constexpr size_t kBoolLds = ((size_t)kBoolPhaseWords + kBoolTileWords) * 4u;

template <int MASK, int FUSED, int KEEP>
static int launch_bool_keep(gl_spmv_plan p, const BoolArgs &a, hipStream_t s) {
    static int attr_device = -1;   // the opt-in is per device (gl_init may switch devices)
    if (attr_device != ctx().device) {
        GL_HIP(hipFuncSetAttribute((const void *)spmv_bool_kernel<MASK, kBoolUnroll, FUSED, KEEP>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBoolLds));
        attr_device = ctx().device;
    }
    Profiler &pf = prof();
    const bool timed = prof_take(pf);
    if (timed) GL_HIP(hipEventRecord(pf.events[2 * pf.used], s));
    spmv_bool_kernel<MASK, kBoolUnroll, FUSED, KEEP><<<p->nunits, kThreads, kBoolLds, s>>>(a);
    GL_LAUNCH_CHECK();
    if (timed) {
        GL_HIP(hipEventRecord(pf.events[2 * pf.used + 1], s));
        pf.used++;
    }
    return GL_OK;
}

template <int MASK, int FUSED>
static int launch_bool_variant(gl_spmv_plan p, const BoolArgs &a, hipStream_t s) {
    static const size_t keep_bytes = (size_t)224 << 20;
    static const size_t keep_min = (size_t)64 << 20;
    // (the CSR copy for the bottom-up BFS step is not streamed by this kernel)
    const size_t rows_bytes = p->d_csr_indptr ? ((size_t)(p->row_end - p->row_begin) + 1u) * 4u + (size_t)p->nnz * 4u : 0u;
    const size_t streamed = p->device_bytes - std::min<size_t>(rows_bytes, p->device_bytes);
    const bool keep = streamed <= keep_bytes && streamed >= keep_min;
    if (p->bool_compressed == 2) return keep ? launch_bool_keep<MASK, FUSED, 7>(p, a, s) : launch_bool_keep<MASK, FUSED, 6>(p, a, s);
    if (p->bool_compressed) return keep ? launch_bool_keep<MASK, FUSED, 3>(p, a, s) : launch_bool_keep<MASK, FUSED, 2>(p, a, s);
    return keep ? launch_bool_keep<MASK, FUSED, 1>(p, a, s) : launch_bool_keep<MASK, FUSED, 0>(p, a, s);
}

template <int MASK>
static int launch_bool(gl_spmv_plan p, const BoolArgs &a, hipStream_t s) {
    return launch_bool_variant<MASK, 0>(p, a, s);
}

// One BFS pull iteration on the bit layout (see the FUSED epilogue).  Unsplit plans whose shard starts on a
// multiple of 64 rows only; bits_out must not alias bits_in.
int bool_plan_bfs_step(gl_spmv_plan p, const uint32_t *bits_in, uint32_t *bits_out, float *d_distance, float level, hipStream_t s,
                       const uint32_t *gate, uint32_t gate_value, int gate_op, uint32_t *back_ctl, uint32_t back_slot,
                       float back_threshold, int back_may_continue, const BfsBitsCtl *v2, const uint32_t *v2_indptr, uint32_t v2_ncols,
                       uint32_t *v2_push_acc, bool v2_deferred) {
    if (p->row_end == p->row_begin) return GL_OK;
    // split plans: only inside the bit-frontier schedule (the claiming epilogue needs an all-zero output vector)
    const bool claim = p->segments > 1 && v2 != nullptr;
    if ((p->segments > 1 && !claim) || (p->row_begin & 63u))
        return set_error(GL_ERR_UNSUPPORTED, "gl_bfs_pull_step: needs an unsplit boolean plan whose shard starts on a multiple of 64 rows");
    if (!p->nunits) return GL_OK;   // no stored entry in this shard: nothing is reached here
    BoolArgs a;
    a.entries = p->d_entries;
    a.bases = p->d_bases;
    a.units = p->d_units;
    a.hub_rows = p->d_hub_rows;
    a.spans = p->d_spans;
    a.xbits = bits_in;
    a.mask = nullptr;
    a.y = nullptr;
    a.zero = 0.0f;
    a.run_flag = nullptr;
    a.bits_out = bits_out;
    a.dist = d_distance;
    a.level = level;
    a.gate = gate;
    a.gate_value = gate_value;
    a.gate_op = gate_op;
    a.back_ctl = back_ctl;
    a.back_slot = back_slot;
    a.back_threshold = back_threshold;
    a.back_may_continue = back_may_continue ? 1u : 0u;
    a.back_n = p->num_rows ? p->num_rows : 1u;
    if (v2) {
        a.v2 = *v2;
        a.v2_indptr = v2_indptr;
        a.v2_ncols = v2_ncols;
        a.v2_push_acc = v2_push_acc;
        a.v2_rowptr = v2_deferred ? nullptr : p->d_csr_indptr;
        a.v2_deferred = v2_deferred;
    }
    a.tickets = bool_tickets();
    if (claim) {
        if (!v2_deferred)
            return set_error(GL_ERR_UNSUPPORTED, "gl_bfs_bits_pull_step: a split plan keeps no totals -- run it with GL_BFS_DEFERRED");
        return launch_bool_variant<GL_NOMASK, 2>(p, a, s);
    }
    return launch_bool_variant<GL_NOMASK, 1>(p, a, s);
}

template <int FUSED, int KEEP>
static int launch_shard_step(uint32_t grid, const BoolArgs &a, const BfsPushArgs &pa, const BfsShardArgs &sa, hipStream_t s) {
    static int attr_device = -1;
    if (attr_device != ctx().device) {
        GL_HIP(hipFuncSetAttribute((const void *)bfs_shard_step_kernel<kBoolUnroll, FUSED, KEEP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kBoolLds));
        attr_device = ctx().device;
    }
    bfs_shard_step_kernel<kBoolUnroll, FUSED, KEEP><<<grid, kThreads, kBoolLds, s>>>(a, pa, sa);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// One slot of the bit-frontier BFS schedule on the rows of `p` in one launch (gl_bfs_bits_shard_step, gl_spmspv.hip fills
// the push arguments from the CSC plan of the same shard).  col_len: global column lengths.
int bool_plan_bfs_shard_step(gl_spmv_plan p, BfsPushArgs pa, BfsShardArgs sa, hipStream_t s) {
    if ((p->row_begin & 63u) || (p->row_end != p->num_rows && (p->row_end & 63u)))
        return set_error(GL_ERR_UNSUPPORTED, "gl_bfs_bits_shard_step: the shard must be cut on multiples of 64 rows");
    BoolArgs a;
    a.entries = p->d_entries;
    a.bases = p->d_bases;
    a.units = p->d_units;
    a.hub_rows = p->d_hub_rows;
    a.spans = p->d_spans;
    a.xbits = pa.bits_in;
    a.mask = nullptr;
    a.y = nullptr;
    a.zero = 0.0f;
    a.run_flag = nullptr;
    a.bits_out = pa.bits_out;
    a.dist = pa.dist;
    a.level = pa.level;
    a.tickets = bool_tickets();
    a.tally_mine = sa.tally_mine;
    a.t_col_len = pa.col_len;
    a.v2_rowptr = p->d_csr_indptr;
    a.v2_row_base = p->row_begin;
    sa.pull_units = p->nunits;
    const uint32_t grid = sa.finish ? 1u : std::max<uint32_t>(std::max<uint32_t>(sa.pull_units, sa.push_blocks), 1u);
    static const size_t keep_bytes = (size_t)224 << 20;
    static const size_t keep_min = (size_t)64 << 20;
    const size_t rows_bytes = p->d_csr_indptr ? ((size_t)(p->row_end - p->row_begin) + 1u) * 4u + (size_t)p->nnz * 4u : 0u;
    const size_t streamed = p->device_bytes - std::min<size_t>(rows_bytes, p->device_bytes);
    const bool keep = streamed <= keep_bytes && streamed >= keep_min;
    if (p->bool_compressed == 2) {
        if (p->segments > 1) return keep ? launch_shard_step<2, 7>(grid, a, pa, sa, s) : launch_shard_step<2, 6>(grid, a, pa, sa, s);
        return keep ? launch_shard_step<1, 7>(grid, a, pa, sa, s) : launch_shard_step<1, 6>(grid, a, pa, sa, s);
    }
    if (p->bool_compressed) {
        if (p->segments > 1) return keep ? launch_shard_step<2, 3>(grid, a, pa, sa, s) : launch_shard_step<2, 2>(grid, a, pa, sa, s);
        return keep ? launch_shard_step<1, 3>(grid, a, pa, sa, s) : launch_shard_step<1, 2>(grid, a, pa, sa, s);
    }
    if (p->segments > 1) return keep ? launch_shard_step<2, 1>(grid, a, pa, sa, s) : launch_shard_step<2, 0>(grid, a, pa, sa, s);
    return keep ? launch_shard_step<1, 1>(grid, a, pa, sa, s) : launch_shard_step<1, 0>(grid, a, pa, sa, s);
}

// d_x != nullptr: pack it into the plan's bit vector first; else run on `bits` as the caller prepared them
// (gl_spmv_run_bits: the bit vector of a row-sharded run is all-gathered instead of the float vector)
int bool_plan_run(gl_spmv_plan p, const float *d_x, const uint32_t *bits, const float *d_mask, float *d_y, float zero,
                  int mask_type, hipStream_t s) {
    const uint32_t rows = p->row_end - p->row_begin;
    if (rows == 0) return GL_OK;
    if (p->segments > 1 || p->nunits == 0) {
        const int rc = spmv_init_rows(GL_OP_ANDOR, mask_type, p->row_begin, p->row_end, d_mask, d_y, zero, s);
        if (rc != GL_OK) return rc;
    }
    if (!p->nunits) return GL_OK;
    if (d_x != nullptr) {
        const uint32_t nwords64 = p->nphases * (kBoolPhaseWords / 2u);
        spmv_bool_pack_kernel<<<std::min<unsigned>(cdiv(nwords64, 4), (unsigned)ctx().num_cus * 16u), 256, 0, s>>>(
            d_x, p->num_cols, reinterpret_cast<uint64_t *>(p->d_xbits), nwords64);
        GL_LAUNCH_CHECK();
        bits = p->d_xbits;
    }
    BoolArgs a;
    a.entries = p->d_entries;
    a.bases = p->d_bases;
    a.units = p->d_units;
    a.hub_rows = p->d_hub_rows;
    a.spans = p->d_spans;
    a.xbits = bits;
    a.mask = d_mask;
    a.y = d_y;
    a.zero = zero;
    a.run_flag = nullptr;
    a.bits_out = nullptr;
    a.dist = nullptr;
    a.level = 0.0f;
    a.tickets = bool_tickets();
    switch (mask_type) {
        case GL_NOMASK: return launch_bool<GL_NOMASK>(p, a, s);
        case GL_MASK_WRITETOZERO: return launch_bool<GL_MASK_WRITETOZERO>(p, a, s);
        case GL_MASK_WRITETOONE: return launch_bool<GL_MASK_WRITETOONE>(p, a, s);
        default: return set_error(GL_ERR_INVALID_ARG, "gl_spmv_run: invalid mask type %d", mask_type);
    }
}

int pack_bits(const float *d_x, uint32_t n, uint32_t *d_bits, hipStream_t s) {
    const uint32_t nwords64 = cdiv(n, 64);
    if (!nwords64) return GL_OK;
    spmv_bool_pack_kernel<<<std::min<unsigned>(cdiv(nwords64, 4), (unsigned)ctx().num_cus * 16u), 256, 0, s>>>(
        d_x, n, reinterpret_cast<uint64_t *>(d_bits), nwords64);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int unpack_bits(const uint32_t *d_bits, uint32_t n, float *d_x, hipStream_t s) {
    if (!n) return GL_OK;
    spmv_bool_unpack_kernel<<<std::min<unsigned>(cdiv(n, 256), (unsigned)ctx().num_cus * 16u), 256, 0, s>>>(d_bits, n, d_x);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int bfs_bits_begin_from(uint32_t *d_ctl, uint32_t ctl_words, const float *d_x, uint32_t n, uint32_t *d_bits, uint32_t words, hipStream_t s,
                        const float *d_distance, gl_spmv_plan rows) {
    const uint32_t words64 = words / 2u;
    const unsigned grid = std::max<unsigned>(cdiv(ctl_words, 256), std::min<unsigned>(cdiv(3u * words64, 4), (unsigned)ctx().num_cus * 16u));
    bfs_bits_begin_from_kernel<<<grid, 256, 0, s>>>(d_ctl, ctl_words, d_x, n, reinterpret_cast<uint64_t *>(d_bits), words64);
    GL_LAUNCH_CHECK();
    if (d_distance && rows && rows->d_csr_indptr && rows->row_begin == 0 && rows->row_end == rows->num_rows && rows->num_rows == n) {
        bfs_bits_visited_kernel<<<std::min<unsigned>(cdiv(n, 256), (unsigned)ctx().num_cus * 8u), 256, 0, s>>>(d_ctl, d_distance, n, rows->d_csr_indptr);
        GL_LAUNCH_CHECK();
    }
    return GL_OK;
}

uint32_t *bool_plan_xbits(gl_spmv_plan p) { return p->d_xbits; }
size_t bool_plan_xbits_bytes(gl_spmv_plan p) { return (size_t)p->nphases * kBoolPhaseWords * 4u; }

// gl_spmspv_run's row-wise path: x bits are already in place, no mask, zero = 0, output rows y[row_begin..row_end);
// the kernels do nothing unless run_flag[0] != 0.  Split plans rely on y being all zero on entry.
int bool_plan_run_bits(gl_spmv_plan p, float *d_y, const uint32_t *run_flag, hipStream_t s, const uint32_t *xbits) {
    if (p->row_end == p->row_begin || !p->nunits) return GL_OK;
    BoolArgs a;
    a.entries = p->d_entries;
    a.bases = p->d_bases;
    a.units = p->d_units;
    a.hub_rows = p->d_hub_rows;
    a.spans = p->d_spans;
    a.xbits = xbits ? xbits : p->d_xbits;
    a.mask = nullptr;
    a.y = d_y;
    a.zero = 0.0f;
    a.run_flag = run_flag;
    a.bits_out = nullptr;
    a.dist = nullptr;
    a.level = 0.0f;
    a.tickets = bool_tickets();
    return launch_bool<GL_NOMASK>(p, a, s);
}

// ------------------------------------------------------------------------------------- planner
// Units of 256*k; tall blocks (the tile is only bits) cut into column segments, because a unit copies every
// x phase it touches into LDS: B blocks x S segments costs about B * (nphases + S - 1) phase copies.
// ---- the delta-coded stream (see the top of the file): one wavefront per group converts it; bad[0] counts the groups that cannot
// be coded (an entry more than 1023 bit positions behind its predecessor, or out of order), bad[1] those with a delta above 255:
// bits 8..9 of every delta go into the two spare bits of its 16-bit row slot, and the kernels of a plan with bad[1] != 0 look there
__global__ __launch_bounds__(256) void bool_compress_kernel(const uint4 *__restrict__ raw, uint32_t ngroups, unsigned char *__restrict__ out,
                                                            uint32_t *__restrict__ bad) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6); g < ngroups; g += gridDim.x * 4u) {
        const uint4 t = raw[(size_t)g * 64u + lane];
        const uint32_t v[4] = {t.x, t.y, t.z, t.w};
        uint32_t idx[4], row[4];
        bool pad[4];
        // a padding entry (ghost row slot) repeats its predecessor's index: the running maximum of the real entries' indices
        uint32_t run = 0u;
        bool sorted = true;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            row[k] = (v[k] >> 5) & kRowPad;
            pad[k] = row[k] == kRowPad;
            idx[k] = ((v[k] >> 19) << 5) | (v[k] & 31u);
            if (!pad[k]) {
                sorted = sorted && idx[k] >= run;
                run = max(run, idx[k]);
            }
        }
        // the lanes in front: inclusive max-scan of the lanes' maxima, shifted by one lane
        uint32_t incl = run;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl = max(incl, up);
        }
        uint32_t before = __shfl_up(incl, 1);
        if (lane == 0) before = 0u;
        uint32_t prev = before, deltas = 0u;
        bool ok = sorted, narrow = true;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t eff = pad[k] ? prev : idx[k];
            const uint32_t dk = eff - prev;
            ok = ok && eff >= prev && dk <= 1023u;
            narrow = narrow && dk <= 255u;
            deltas |= (dk & 255u) << (8 * k);
            row[k] |= ((dk >> 8) & 3u) << kRowBits;
            prev = eff;
        }
        if (__ballot(!ok) != 0ull) {
            if (lane == 0) atomicAdd(bad, 1u);
            continue;
        }
        if (__ballot(!narrow) != 0ull && lane == 0) atomicAdd(bad + 1, 1u);
        uint32_t *o = reinterpret_cast<uint32_t *>(out + (size_t)g * kBoolGroupBytesC) + 3u * lane;
        o[0] = row[0] | (row[1] << 16), o[1] = row[2] | (row[3] << 16), o[2] = deltas;
    }
}

// after either formatter: re-code the plan's groups to 3 bytes per entry if every one of them allows it (else the plan stays as it is)
int bool_plan_compress(gl_spmv_plan p) {
    if (!p->boolean || kBoolLane != 4 || p->ngroups == 0 || debug_knob("bool_compress", 1) == 0) return GL_OK;
    hipStream_t s = ctx().stream;
    const size_t bytes = (size_t)p->ngroups * kBoolGroupBytesC;
    unsigned char *d_new = nullptr;
    uint32_t *d_bad = nullptr;
    GL_HIP(hipMalloc((void **)&d_new, bytes));
    if (hipMalloc((void **)&d_bad, 16) != hipSuccess) {
        (void)hipFree(d_new);
        return set_error(GL_ERR_HIP, "bool_plan_compress: out of device memory");
    }
    hipError_t e = hipMemsetAsync(d_bad, 0, 16, s);
    if (e == hipSuccess) {
        bool_compress_kernel<<<std::min<uint32_t>((uint32_t)cdiv((uint32_t)p->ngroups, 4u), (uint32_t)ctx().num_cus * 16u), 256, 0, s>>>(
            reinterpret_cast<const uint4 *>(p->d_entries), (uint32_t)p->ngroups, d_new, d_bad);
        e = hipGetLastError();
    }
    uint32_t bad = 1u, wide = 0u;
    if (e == hipSuccess) e = d2h_word_sync(&bad, d_bad, s);
    if (e == hipSuccess) e = d2h_word_sync(&wide, d_bad + 1, s);
    (void)hipFree(d_bad);
    if (e != hipSuccess) {
        (void)hipFree(d_new);
        return set_error(GL_ERR_HIP, "bool_plan_compress: %s", hipGetErrorString(e));
    }
    const long knob = debug_knob("bool_compress", 1);
    if (knob == 2)
        fprintf(stderr, "bool_plan_compress: of %llu groups %u cannot be delta-coded, %u need 10-bit deltas\n", (unsigned long long)p->ngroups, bad, wide);
    // some group has a gap of more than 1023 columns: the 4-byte stream stays.  So it does when the 10-bit decoder would be needed
    // on a stream that the Infinity Cache holds anyway (pokec, 128 MB: 0.033 ms as it is, 0.036 coded -- nothing to save but VALU
    // work added; the 8-bit decoder is even there, googleplus, and ahead from ogbl_ppa's 160 MB on)
    if (bad || (wide && knob != 3 && p->b_entries <= ((size_t)224 << 20))) {
        (void)hipFree(d_new);
        return GL_OK;
    }
    (void)hipFree(p->d_entries);
    p->d_entries = reinterpret_cast<uint2 *>(d_new);
    p->device_bytes -= p->b_entries;
    p->b_entries = bytes;
    p->device_bytes += bytes;
    p->bool_compressed = (wide || knob == 3) ? 2 : 1;   // (knob 3: the 10-bit decoder on a plan that does not need it -- what it costs)
    return GL_OK;
}

static Shape choose_shape_bool(uint64_t rows, uint64_t cols, uint64_t nnz, int num_cus) {
    Shape best{1, 1};
    if (rows == 0 || nnz == 0) return best;
    const double nph = (double)cdiv(cols, kBoolPhaseCols);
    const double phase_bytes = std::min<double>((double)kBoolPhaseWords * 4.0, (double)cols / 8.0);
    const uint64_t rmax = kBoolHubBit0 - 64;
    double best_cost = 1e300;
    for (int k = 1; k <= 16 && best_cost > 1e299; k *= 2) {
        for (uint32_t S = 1; S <= 64; S++) {
            uint64_t B = (uint64_t)num_cus * k / S;
            if (B == 0) break;
            if (B > rows) B = rows;
            const uint64_t R = (rows + B - 1) / B;
            if (R > rmax) continue;
            const double util = (double)B * S / ((double)num_cus * k);
            const double t_stream = 4.0 * (double)nnz / 6.5e12 / util;
            const double copies = (S == 1) ? nph : (nph + S - 1) / S;          // per unit
            const double t_bits = k * copies * phase_bytes / 45e9;              // one CU copies ~45 GB/s from L2
            const double t_fold = (S == 1) ? 0.0 : (double)rows * 8.0 / 4e12 + 3e-6;
            // the BFS epilogues walk a unit's rows 4096 at a time, each step a chain of dependent round trips (measured on
            // one emulated rank of 8 of the orkut stand-in: 32 x 8 units 100 us per BFS, 64 x 4 93 us, 128 x 2 94 us, 256 x 1 98 us)
            const double t_epi = (double)cdiv(R, 4096) * (S == 1 ? 4.0e-6 : 2.5e-6);
            const double t = t_stream + t_bits + t_fold + t_epi + 3.0e-6 * k;
            if (t < best_cost) {
                best_cost = t;
                best = Shape{(uint32_t)B, S};
            }
        }
    }
    if (best_cost > 1e299) best = Shape{(uint32_t)((rows + rmax - 1) / rmax), 1};
    const long fb = debug_knob("spmv_blocks", 0), fs = debug_knob("spmv_segments", 0);
    if (fb > 0) best.blocks = (uint32_t)std::min<uint64_t>((uint64_t)fb, rows);
    if (fs > 0) best.segments = (uint32_t)std::min<long>(fs, 4096);
    return best;
}

int bool_plan_build(gl_spmv_plan p, const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data) {
    const uint32_t num_cols = p->num_cols, row_begin = p->row_begin, row_end = p->row_end;
    const uint32_t rows = row_end - row_begin;
    const Shape shape = choose_shape_bool(rows, num_cols, p->nnz, ctx().num_cus);
    // block boundaries on multiples of 64 rows: the fused BFS epilogue writes whole 64-bit frontier words
    const BlockPlan bp = plan_blocks(shape, h_indptr, row_begin, row_end, kBoolHubBit0 / 64u * 64u, 64u);
    const uint32_t nblocks = bp.nblocks, nunits = bp.nunits;
    const uint32_t nphases = cdiv(num_cols, kBoolPhaseCols);

    if (format_on_device(p->flags, p->nnz)) {
        // the record loop below as kernels over a staged copy of the shard (gl_format.hip); identical arrays
        struct Staged {
            DevCsr *c = nullptr;
            ~Staged() { devcsr_release(c); }
        } staged;
        int rc = devcsr_stage(&staged.c, h_indptr, h_indices, h_data, row_begin, row_end);
        if (rc != GL_OK) return rc;
        EmitBool eb;
        eb.bp = &bp;
        eb.h_indptr = h_indptr;
        eb.num_cols = num_cols;
        uint32_t tallest = 0;
        if ((rc = fmt_emit_bool(staged.c, eb, p, &tallest)) != GL_OK) return rc;
        if (debug_knob("bfs_keep_rows", 1) != 0) {
            if ((rc = devcsr_adopt_rows(staged.c, &p->d_csr_indptr, &p->d_csr_indices)) != GL_OK) return rc;
            p->csr_nz_base = h_indptr[row_begin];
            p->device_bytes += ((size_t)rows + 1u) * 4u + (size_t)p->nnz * 4u;
        }
        p->boolean = true;
        p->nblocks = nblocks;
        p->segments = bp.Smax;
        p->nunits = nunits;
        p->max_block_rows = tallest;
        p->nphases = nphases;
        GL_HIP(hipMalloc((void **)&p->d_xbits, (size_t)nphases * kBoolPhaseWords * 4u));
        p->device_bytes += (size_t)nphases * kBoolPhaseWords * 4u;
        return GL_OK;
    }

    // every unit is emitted on its own (groups, bases, spans), then the pieces are laid out in unit order
    struct UnitOut {
        std::vector<uint32_t> ent;     // 128 per group
        std::vector<uint32_t> bases;   // per group
        std::vector<uint4> spans;      // group indices are unit-local until the final pass
    };
    std::vector<UnitOut> out(nunits);
    std::vector<uint4> units((size_t)nunits * 2);
    std::vector<uint32_t> hub_rows((size_t)nblocks * kBoolHubMax, 0u);
    int bad_col = 0;
    uint32_t max_rows = 0;
#pragma omp parallel
    {
        std::vector<Rec> recs, tmp;
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < (int64_t)nblocks; b++) {
            const uint32_t r0 = bp.bstart[b], r1 = bp.bstart[b + 1];
            recs.clear();
            bool bad = false;
            for (uint32_t r = r0; r < r1; r++)
                for (uint64_t i = h_indptr[r]; i < h_indptr[r + 1]; i++) {
                    const uint32_t c = h_indices[i];
                    if (c >= num_cols) { bad = true; continue; }
                    if (h_data[i] != 0.0f) recs.push_back(Rec{c, r - r0, 0u});   // a && b is false for a == 0
                }
            if (bad) {
#pragma omp atomic write
                bad_col = 1;
                continue;
            }
            sort_by_col(recs, tmp, num_cols);
            const uint64_t m = recs.size();
            // hub rows: rows that own a large share of the block's entries receive most of the hits of a BFS
            // step; their bit is spread over 32 private bits in 32 different tile words (picked by the entry's
            // position in its group) so that the ds_or of one wavefront step do not pile up on one word
            std::vector<uint32_t> cnt(r1 - r0, 0u);
            for (const Rec &rc : recs) cnt[rc.row_local]++;
            std::vector<int> hub_of(r1 - r0, -1);
            uint32_t nhub = 0;
            {
                const uint64_t thr = std::max<uint64_t>(256, m / 48);
                for (uint32_t i = 0; i < r1 - r0 && nhub < kBoolHubMax; i++)
                    if (cnt[i] >= thr) {
                        hub_of[i] = (int)nhub;
                        hub_rows[(size_t)b * kBoolHubMax + nhub] = i;
                        nhub++;
                    }
            }
            auto slot_of = [&](const Rec &rc, uint32_t fill) -> uint32_t {
                const int hb = hub_of[rc.row_local];
                return hb < 0 ? rc.row_local : kBoolHubBit0 + (fill & (kBoolHubSlots - 1u)) * 32u + (uint32_t)hb;
            };
            const uint32_t S = bp.seg[b];
            for (uint32_t s = 0; s < S; s++) {
                const size_t u = bp.unit_of[s][b];
                UnitOut &o = out[u];
                uint32_t fill = 0, base = 0, phase = 0xffffffffu, lo = 0, hi = 0;   // hi: column of the last entry
                bool open = false;
                auto seal_group = [&]() {   // pad the open group with entries of the ghost row slot, column = base
                    if (!open) return;
                    for (; fill < kBoolGroup; fill++) o.ent.push_back(kRowPad << 5);
                    open = false;
                };
                auto close_span = [&]() {
                    if (phase == 0xffffffffu) return;
                    seal_group();
                    uint4 &sp = o.spans.back();
                    sp.z = (uint32_t)o.bases.size();
                    sp.w = (lo / 128u) | (((hi / 128u) + 1u) << 16);   // uint4 (128-column) granules of the phase
                };
                for (uint64_t i = m * s / S; i < m * (s + 1) / S; i++) {
                    const Rec &rc = recs[i];
                    const uint32_t ph = rc.col / kBoolPhaseCols, cin = rc.col - ph * kBoolPhaseCols;
                    if (ph != phase) {
                        close_span();
                        phase = ph;
                        o.spans.push_back(make_uint4(ph * kBoolPhaseWords, (uint32_t)o.bases.size(), 0u, 0u));
                        lo = cin;
                    }
                    if (!open || fill == kBoolGroup || cin - base >= (1u << kColOffBits)) {
                        seal_group();
                        base = cin & ~31u;                 // groups start on an x word
                        o.bases.push_back(base >> 5);
                        fill = 0;
                        open = true;
                    }
                    o.ent.push_back((((cin - base) >> 5) << 19) | (slot_of(rc, fill) << 5) | (cin & 31u));
                    fill++;
                    hi = cin;
                }
                close_span();
                units[2 * u] = make_uint4(0u, (uint32_t)o.spans.size(), r0, (r1 - r0) | (bp.all_direct ? 0x80000000u : 0u));
                units[2 * u + 1] = make_uint4((uint32_t)((size_t)b * kBoolHubMax), nhub, 0u, 0u);
            }
        }
    }
    if (bad_col)
        return set_error(GL_ERR_INVALID_ARG, "gl_spmv_plan_create: column index out of range (num_cols %u)", num_cols);
    for (uint32_t b = 0; b < nblocks; b++) max_rows = std::max(max_rows, bp.bstart[b + 1] - bp.bstart[b]);

    std::vector<uint64_t> goff((size_t)nunits + 1, 0), soff((size_t)nunits + 1, 0);
    for (size_t u = 0; u < nunits; u++) {
        goff[u + 1] = goff[u] + out[u].bases.size();
        soff[u + 1] = soff[u] + out[u].spans.size();
    }
    const uint64_t total_groups = goff[nunits];
    if (total_groups >= 0xffffffffull) return set_error(GL_ERR_UNSUPPORTED, "gl_spmv_plan_create: too many groups");
    std::vector<uint32_t> entries(total_groups * kBoolGroup), bases(total_groups);
    std::vector<uint4> spans(soff[nunits]);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t u = 0; u < (int64_t)nunits; u++) {
        UnitOut &o = out[u];
        if (!o.bases.empty()) {
            memcpy(&entries[goff[u] * kBoolGroup], o.ent.data(), o.ent.size() * 4u);
            memcpy(&bases[goff[u]], o.bases.data(), o.bases.size() * 4u);
        }
        for (size_t k = 0; k < o.spans.size(); k++) {
            uint4 sp = o.spans[k];
            sp.y += (uint32_t)goff[u];
            sp.z += (uint32_t)goff[u];
            spans[soff[u] + k] = sp;
        }
        units[2 * u].x = (uint32_t)soff[u];
        UnitOut().ent.swap(o.ent);
    }

    p->boolean = true;
    p->nblocks = nblocks;
    p->segments = bp.Smax;
    p->nunits = nunits;
    p->ngroups = total_groups;
    p->max_block_rows = max_rows;
    p->nphases = nphases;
    auto up = [&](void **d, const void *h, size_t bytes) -> int {
        GL_HIP(hipMalloc(d, bytes ? bytes : 16));
        if (bytes) GL_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
        p->device_bytes += bytes;
        return GL_OK;
    };
    int rc;
    if ((rc = up((void **)&p->d_entries, entries.data(), entries.size() * 4u)) != GL_OK ||
        (rc = up((void **)&p->d_bases, bases.data(), bases.size() * 4u)) != GL_OK ||
        (rc = up((void **)&p->d_units, units.data(), units.size() * sizeof(uint4))) != GL_OK ||
        (rc = up((void **)&p->d_hub_rows, hub_rows.data(), hub_rows.size() * sizeof(uint32_t))) != GL_OK ||
        (rc = up((void **)&p->d_spans, spans.data(), spans.size() * sizeof(uint4))) != GL_OK)
        return rc;
    p->b_entries = entries.size() * 4u;
    p->b_bases = bases.size() * 4u;
    p->b_units = units.size() * sizeof(uint4);
    p->b_hub_rows = hub_rows.size() * sizeof(uint32_t);
    p->b_spans = spans.size() * sizeof(uint4);
    GL_HIP(hipMalloc((void **)&p->d_xbits, (size_t)nphases * kBoolPhaseWords * 4u));
    p->device_bytes += (size_t)nphases * kBoolPhaseWords * 4u;
    if (debug_knob("bfs_keep_rows", 1) != 0) {
        // the rows as plain CSR for the bottom-up BFS step (see gl_spmv_plan.h); zero values -> column 0xffffffff
        const uint64_t nz0 = h_indptr[row_begin];
        std::vector<uint32_t> cols(h_indices + nz0, h_indices + nz0 + p->nnz);
        for (uint64_t i = 0; i < p->nnz; i++)
            if (h_data[nz0 + i] == 0.0f) cols[i] = 0xffffffffu;
        if ((rc = up((void **)&p->d_csr_indptr, h_indptr + row_begin, ((size_t)rows + 1u) * 4u)) != GL_OK ||
            (rc = up((void **)&p->d_csr_indices, cols.data(), cols.size() * 4u)) != GL_OK)
            return rc;
        p->csr_nz_base = (uint32_t)nz0;
    }
    return GL_OK;
}

}  // namespace gl

// gl_init loads this translation unit's code object up front (HIP defers that to the unit's first launch, which would put
// tens of milliseconds into somebody's timed call)
namespace gl {
int preload_spmv_bool() {
    hipFuncAttributes attr;
    GL_HIP(hipFuncGetAttributes(&attr, (const void *)spmv_bool_pack_kernel));
    return GL_OK;
}
}  // namespace gl
