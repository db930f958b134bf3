// The bit-frontier BFS schedule on a ROW SHARD in ONE launch per slot (gl_bfs_bits_shard_step).
//
// The two-launch slot of gl_spmspv.hip / gl_spmv_bool.hip (push step, pull step) plus gl_bfs_bits_decide on the gathered
// vector costs a shard three dependent launches per slot, and a 1/8 shard of the orkut stand-in has less work per slot than
// one launch costs.  Here a slot is one launch:
//   * every rank TALLIES what its own step reached -- {vertices, their global column lengths, their row lengths}, the three
//     numbers BfsBitsCtl::decide goes by -- into kTallyLines lines of 32 bytes, which travel with the slot's bit-vector
//     all-gather (256 bytes per rank);
//   * the launch of slot s begins with the decision of slot s - 1: EVERY workgroup adds up all ranks' lines (integers: the
//     same sums on every rank and in every workgroup) and replays decide() on a private copy of the 16 control words in LDS.
//     Workgroup 0 stores that state for the next launch into the OTHER of two state buffers (the one nobody reads in this
//     launch) and writes the per-slot records the host reads;
//   * then the workgroups run the slot's step as the state says: the scattering push, the bottom-up scan or the streaming
//     pull -- the same bodies as the two-launch slot's, in one kernel of 1024-thread workgroups.
// The reference's loop condition (app/bfs.h:180-190) is evaluated exactly as in the one-GPU schedule: decide() is the same
// function on the same three totals.
#ifndef GL_BFS_SHARD_H_
#define GL_BFS_SHARD_H_

#include "gl_common.h"

namespace gl {

#ifndef GL_BFS_CHUNK
#define GL_BFS_CHUNK 1024
#endif
constexpr uint32_t kBfsChunk = GL_BFS_CHUNK;   // bit-frontier BFS push step: columns at least this long are served from the plan's
                                               // static list of chunks of this many entries (two round trips of a workgroup)

// ------------------------------------------------------------------ BFS push step on a bit frontier (gl_bfs_bits_push_step)
// SpMSpV (||,&&) masked WriteToZero by the distances + AssignVectorSparse(level) (app/bfs.h:146-148) with the bit vector
// of the next frontier as the accumulator: a product whose row is still unvisited sets the row's bit, and the thread that
// sets it first writes the level -- no dense accumulator, no compaction, ONE launch.  Long columns are not queued at run
// time: the plan lists their chunks, every workgroup tests the frontier bit of the chunks it is dealt.
struct BfsPushArgs {
    const uint32_t *indptr;
    const uint2 *stream;
    const uint4 *chunks;
    uint32_t nchunks;
    uint32_t num_cols;
    const uint32_t *bits_in;
    uint32_t *bits_out;      // all zero on entry (the push step of two slots earlier cleared it)
    uint32_t *bits_spare;    // cleared here, gate or not: the next slot's bits_out
    uint32_t words;          // words of each bit vector
    uint32_t col_words;      // words that hold columns
    float *dist;
    float level;
    uint32_t *acc;           // kBfsAccSlots x 32 words, zero between steps: [0] new vertices, [2..3] their column lengths,
                             // [6..7] their row lengths ([4]: the pull step's per-line ticket)
    const uint32_t *row_ptr; // the rows as plain CSR (boolean SpMV plan of the same matrix and shard), or null:
    const uint32_t *row_idx; //   row lengths for the bookkeeping, and the bottom-up branch
    uint32_t num_rows;
    // row shard [row_begin, row_end): row_ptr is indexed by row - row_begin and holds GLOBAL offsets, row_idx is indexed by
    // offset - nz_base.  deferred: the step keeps no totals and takes no decision -- a shard's counts are partial; the driver
    // all-gathers the next frontier and runs gl_bfs_bits_decide on it
    uint32_t row_begin, row_end, nz_base;
    bool deferred;
    // one-launch shard step: the column lengths of the WHOLE matrix (a shard's own column pointers count its rows only)
    const uint32_t *col_len = nullptr;
    // one-launch shard step, scattering push: frontier bits a lane takes (a power of two, 1 ... 32): a wavefront's strip is
    // 64 x bpl bits.  Small graphs get narrow strips so that every wavefront of the grid has one (see bfs_shard_scatter)
    uint32_t bpl = 32;
    BfsBitsCtl c;
};

// the first thread to set an unvisited row's bit writes its level and counts it (and the row's column: the next push's work)
__device__ __forceinline__ void bfs_claim(const BfsPushArgs &a, bool cand, uint32_t row, uint32_t &fresh, uint32_t &work, uint32_t &work_rows) {
    if (!cand) return;
    const uint32_t m = 1u << (row & 31u);
    const uint32_t old = atomicOr(&a.bits_out[row >> 5], m);
    if (old & m) return;
    a.dist[row] = a.level;
    if (a.deferred) return;
    fresh += 1u;
    if (a.col_len) work += a.col_len[row];
    else if (row < a.num_cols) work += a.indptr[row + 1u] - a.indptr[row];
    if (a.row_ptr) work_rows += a.row_ptr[row - a.row_begin + 1u] - a.row_ptr[row - a.row_begin];
}
// candidate = the product a && x is true and the mask (distance == 0: not visited, app/bfs.h:146) lets it through
__device__ __forceinline__ bool bfs_candidate(const BfsPushArgs &a, bool valid, uint2 rv) {
    return valid && (rv.y << 1) != 0u && a.dist[rv.x] == 0.0f;
}

// ------------------------------------------------------------------ tallies
// d_tally of gl_bfs_bits_shard_step: kTallyHeadWords words (the second state buffer), then per slot (1-based) and rank
// kTallyLines lines of kTallyLineWords words: [0] vertices reached, [2..3] their global column lengths, [4..5] their row lengths
constexpr uint32_t kTallyLines = 8, kTallyLineWords = 8, kTallyRankWords = kTallyLines * kTallyLineWords, kTallyHeadWords = 64;

// The streamed read-back of a BFS result (gl_levels_pack_stream, include/graphlily_hip.h): levels -> nibbles / bytes stored straight
// into a page-locked host block, chunk by chunk, a flag word raised behind every chunk (system-scope release: the flag does not
// overtake the data on its way to the host) and behind the tail words -- the host expands chunk k while chunk k + 1 crosses PCIe.
struct LevelsPack {
    const float4 *src = nullptr;
    uint32_t *dst = nullptr;       // the block, as the device sees it; null: nothing to pack
    uint32_t *flags = nullptr;     // nchunks flags, GL_LEVELS_FLAG_STRIDE_WORDS apart; the last one is the tail's
    const uint32_t *tail = nullptr;
    uint32_t nwords = 0, bits = 4, tail_words = 0, tail_at = 0, nchunks = 0;
};

template <int BITS>
__device__ __forceinline__ uint32_t levels_pack_word(const float4 *__restrict__ src, uint32_t i) {
    constexpr uint32_t M = (1u << BITS) - 1u;
    if (BITS == 8) {
        const float4 v = src[i];
        return ((uint32_t)v.x & M) | (((uint32_t)v.y & M) << 8) | (((uint32_t)v.z & M) << 16) | (((uint32_t)v.w & M) << 24);
    }
    const float4 v = src[2u * i], w = src[2u * i + 1u];
    return ((uint32_t)v.x & M) | (((uint32_t)v.y & M) << 4) | (((uint32_t)v.z & M) << 8) | (((uint32_t)v.w & M) << 12) |
           (((uint32_t)w.x & M) << 16) | (((uint32_t)w.y & M) << 20) | (((uint32_t)w.z & M) << 24) | (((uint32_t)w.w & M) << 28);
}

// the whole workgroup: raise flag c once every thread's stores have left for the host.  ONE system-scope release per workgroup: every
// wavefront waits until its stores have reached the L2 (s_waitcnt vmcnt(0): the compiler's workgroup-scope release waits for nothing,
// one L1 serves the workgroup), the barrier, then thread 0's release -- the write-back of the L2 (the block's lines do sit there:
// with the wait alone and a relaxed flag store, stale chunks reached the host behind their flags) and the flag behind it.
// __threadfence_system() in every thread was a write-back walk of the L2 per WAVEFRONT: 756 of them made this pack 42 us long
// instead of 31 and delayed its start by 34 us in every replay of the recorded schedule (profiles/r06_bfs_trace.txt).
__device__ __forceinline__ void levels_pack_flag(const LevelsPack &k, uint32_t c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(k.flags + (size_t)c * GL_LEVELS_FLAG_STRIDE_WORDS, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// chunk c (< nchunks - 1) by the whole workgroup: 16 bytes per lane and store (1 KB per wavefront instruction: whole PCIe write
// bursts), the odd words of the last chunk singly
template <int BITS>
__device__ __forceinline__ void levels_pack_chunk(const LevelsPack &k, uint32_t c) {
    const uint32_t begin = c * GL_LEVELS_CHUNK_WORDS, end = min(k.nwords, begin + GL_LEVELS_CHUNK_WORDS), end4 = begin + ((end - begin) & ~3u);
    for (uint32_t i = begin + 4u * threadIdx.x; i < end4; i += 4u * blockDim.x)
        *reinterpret_cast<uint4 *>(k.dst + i) = make_uint4(levels_pack_word<BITS>(k.src, i), levels_pack_word<BITS>(k.src, i + 1u),
                                                           levels_pack_word<BITS>(k.src, i + 2u), levels_pack_word<BITS>(k.src, i + 3u));
    for (uint32_t i = end4 + threadIdx.x; i < end; i += blockDim.x) k.dst[i] = levels_pack_word<BITS>(k.src, i);
    levels_pack_flag(k, c);
}

// the tail words (the schedule's control words), by the whole workgroup
__device__ __forceinline__ void levels_pack_tail(const LevelsPack &k) {
    for (uint32_t i = threadIdx.x; i < k.tail_words; i += blockDim.x) k.dst[k.tail_at + i] = k.tail[i];
    levels_pack_flag(k, k.nchunks - 1u);
}

// gl_runtime.hip: checks the block (page-locked, device-visible, aligned) and fills the descriptor for n levels + tail_words words
int levels_stream_describe(const float *d_levels, uint32_t n, int bits, const uint32_t *d_tail, uint32_t tail_words, void *h_block,
                           LevelsPack *out, const char *who);

struct BfsShardArgs {
    const uint32_t *state_in;    // 16 control words: the state the previous launch stored (gl_bfs_bits_begin's in slots 1 and 2)
    uint32_t *state_out;         // workgroup 0 stores the state after slot - 1 here (null in slot 1: nothing to decide yet)
    uint32_t *records;           // the caller's control words: ctl[15] and the per-slot records behind the 16 words
    const uint32_t *tally_prev;  // every rank's lines of slot - 1, after the exchange (null in slot 1)
    uint32_t lines_prev;         // world x kTallyLines
    uint32_t *tally_mine;        // this rank's lines of this slot (zero on entry)
    BfsBitsCtl prev;             // the parameters decide(slot - 1) runs with
    uint32_t slot;
    uint32_t pull_units;         // workgroups the streaming pull needs
    uint32_t push_blocks;        // workgroups the scattering push uses
    uint32_t finish;             // 1: only take the last slot's decision (gl_bfs_bits_shard_finish)
};

// a workgroup's totals -> one of the rank's lines (all threads call; per-lane counts)
__device__ __forceinline__ void tally_block_add(uint32_t *mine, uint32_t blk, uint32_t fresh, uint32_t work, uint32_t rows) {
    __shared__ uint32_t t_fresh;
    __shared__ unsigned long long t_work, t_rows;
    if (threadIdx.x == 0) {
        t_fresh = 0u;
        t_work = 0ull;
        t_rows = 0ull;
    }
    __syncthreads();
    unsigned long long w64 = work, r64 = rows;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        fresh += __shfl_down(fresh, d);
        w64 += __shfl_down(w64, d);
        r64 += __shfl_down(r64, d);
    }
    if ((threadIdx.x & 63u) == 0u && fresh) {
        atomicAdd(&t_fresh, fresh);
        atomicAdd(&t_work, w64);
        atomicAdd(&t_rows, r64);
    }
    __syncthreads();
    if (threadIdx.x == 0 && t_fresh) {
        uint32_t *line = mine + kTallyLineWords * (blk & (kTallyLines - 1u));
        atomicAdd(line, t_fresh);
        atomicAdd(reinterpret_cast<unsigned long long *>(line + 2), t_work);
        if (t_rows) atomicAdd(reinterpret_cast<unsigned long long *>(line + 4), t_rows);
    }
}

// The launch's first act: the decision of slot - 1 on a private copy of the control words (s_state: 16 words of LDS).
// Returns the view of the state the slot's step goes by.
__device__ __forceinline__ BfsBitsCtl shard_prologue(const BfsShardArgs &sa, uint32_t *s_state) {
    if (threadIdx.x < 16u) s_state[threadIdx.x] = sa.state_in[threadIdx.x];
    uint32_t f = 0u;
    unsigned long long w = 0ull, r = 0ull;
    if (threadIdx.x < 64u && sa.tally_prev) {
        for (uint32_t i = threadIdx.x; i < sa.lines_prev; i += 64u) {
            const uint32_t *ln = sa.tally_prev + kTallyLineWords * i;
            f += ln[0];
            w += *reinterpret_cast<const unsigned long long *>(ln + 2);
            r += *reinterpret_cast<const unsigned long long *>(ln + 4);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            f += __shfl_down(f, d);
            w += __shfl_down(w, d);
            r += __shfl_down(r, d);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && sa.tally_prev) {
        BfsBitsCtl p = sa.prev;
        p.ctl = s_state;
        p.local = true;
        p.records = blockIdx.x == 0 ? sa.records : nullptr;
        p.decide(f, w, r);
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 16u && sa.state_out) sa.state_out[threadIdx.x] = s_state[threadIdx.x];
    BfsBitsCtl c = sa.prev;
    c.ctl = s_state;
    c.local = true;
    c.records = blockIdx.x == 0 ? sa.records : nullptr;
    c.slot = sa.slot;
    return c;
}

// The scattering push for workgroups that cannot be many (the streaming pull's LDS tile leaves room for ONE workgroup of T
// threads per compute unit): every WAVEFRONT works on its own -- no workgroup barrier in the main loop, 16 independent
// latency chains per compute unit.  A wavefront takes a strip of 64 x bpl frontier bits (bpl per lane); per round every lane
// with bits left takes its lowest one = up to 64 columns, whose entries are dealt to the lanes in order (prefix of the column
// lengths in the wavefront's 128 words of LDS, binary search per entry): two entries per lane and step in flight.
// A round is a chain of four dependent round trips (column pointers, entries, distances, claims) and a lane has up to bpl
// rounds: with whole words per lane (bpl = 32) the googleplus stand-in's 3375 frontier words were 53 strips -- 53 busy
// wavefronts on the whole chip -- and a 20 762-vertex frontier of short columns took 12 rounds = 41 us (round 4,
// profiles/r04_bfs_timeline_googleplus.txt); the host now picks bpl so that the strips cover the grid's wavefronts.
// Columns of kBfsChunk entries and more come from the plan's chunk list, by the whole workgroup (as in gl_spmspv.hip).
// `lds`: >= 128 * (T / 64) + T + 8 + 512 words.  Workgroup `blk` of `G`.
template <uint32_t T>
__device__ __forceinline__ void bfs_shard_scatter(const BfsPushArgs &a, uint32_t *lds, uint32_t blk, uint32_t G, uint32_t &fresh,
                                                  uint32_t &work, uint32_t &work_rows) {
    constexpr uint32_t W = T / 64u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *s_pref = lds + 128u * wave, *s_first = s_pref + 64u;   // this wavefront's: exclusive prefix of the lengths, first entries
    uint32_t *s_hits = lds + 128u * W, *s_nhit = s_hits + T;
    // columns too long for one wavefront (more than kWaveMax entries, fewer than a chunk) are set aside -- {first entry,
    // length} in s_long -- and applied by the whole workgroup behind the strips: a hub vertex in a light frontier is a
    // serial chain of round trips for one wavefront (an 800-entry column: 7 steps, 30 us) and one step for 1024 threads
    constexpr uint32_t kWaveMax = 128u, kLongCap = 256u;
    uint32_t *s_long = s_nhit + 4u, *s_nlong = s_nhit + 1u;
    if (tid == 0) {
        *s_nhit = 0u;
        *s_nlong = 0u;
    }
    __syncthreads();
    const uint32_t bpl = a.bpl, strip_bits = 64u * bpl;
    const uint32_t nstrips = (a.col_words * 32u + strip_bits - 1u) / strip_bits;
    for (uint32_t strip = blk + G * wave; strip < nstrips; strip += G * W) {   // (neighbouring strips go to different compute units)
        const uint32_t bit0 = strip * strip_bits + lane * bpl;      // this lane's first frontier bit
        const uint32_t wi = bit0 >> 5;
        uint32_t w = wi < a.col_words ? a.bits_in[wi] : 0u;
        if (bpl < 32u) w = (w >> (bit0 & 31u)) & ((1u << bpl) - 1u);
        while (__any(w != 0u)) {
            uint32_t start = 0u, deg = 0u;
            if (w) {
                const uint32_t col = bit0 + (uint32_t)__ffs((int)w) - 1u;
                w &= w - 1u;
                if (col < a.num_cols) {
                    start = a.indptr[col];
                    deg = a.indptr[col + 1u] - start;
                    if (deg >= kBfsChunk) deg = 0u;       // served from the chunk list below
                    else if (deg > kWaveMax) {
                        const uint32_t at = atomicAdd(s_nlong, 1u);
                        if (at < kLongCap) {              // (a full list: the wavefront applies the column itself)
                            s_long[2u * at] = start;
                            s_long[2u * at + 1u] = deg;
                            deg = 0u;
                        }
                    }
                }
            }
            uint32_t incl = deg;
#pragma unroll
            for (uint32_t dlt = 1; dlt < 64; dlt <<= 1) {
                const uint32_t up = __shfl_up(incl, dlt);
                if (lane >= dlt) incl += up;
            }
            const uint32_t total = __shfl(incl, 63);
            if (!total) continue;
            s_pref[lane] = incl - deg;
            s_first[lane] = start;
            for (uint32_t e0 = 0; e0 < total; e0 += 128u) {
                uint2 rv[2];
                bool valid[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t idx = e0 + 64u * u + lane;
                    valid[u] = idx < total;
                    uint32_t lo = 0u, hi = 63u;   // largest j with s_pref[j] <= idx (empty columns share a prefix: the largest owns the entry)
#pragma unroll
                    for (int it = 0; it < 6; it++) {
                        const uint32_t mid = (lo + hi + 1u) >> 1;
                        if (s_pref[mid] <= idx) lo = mid; else hi = mid - 1u;
                    }
                    rv[u] = valid[u] ? load_stream_nt(a.stream + s_first[lo] + (idx - s_pref[lo])) : make_uint2(0u, 0u);
                }
                const bool c0 = bfs_candidate(a, valid[0], rv[0]), c1 = bfs_candidate(a, valid[1], rv[1]);
                bfs_claim(a, c0, rv[0].x, fresh, work, work_rows);
                bfs_claim(a, c1, rv[1].x, fresh, work, work_rows);
            }
        }
    }
    __syncthreads();
    {
        const uint32_t nlong = min(*s_nlong, kLongCap);
        for (uint32_t h = 0; h < nlong; h++) {
            const uint32_t start = s_long[2u * h], deg = s_long[2u * h + 1u];
            for (uint32_t k = tid; k < deg; k += T) {
                const uint2 r0 = load_stream_nt(a.stream + start + k);
                bfs_claim(a, bfs_candidate(a, true, r0), r0.x, fresh, work, work_rows);
            }
        }
    }
    if (!a.nchunks) return;
    // chunks of long columns: every thread tests the frontier bit of one chunk, the hits are processed by the whole workgroup
    for (uint32_t q0 = 0; q0 < a.nchunks; q0 += G * T) {
        const uint32_t q = q0 + tid * G + blk;
        if (q < a.nchunks) {
            const uint32_t col = a.chunks[q].x;
            if ((a.bits_in[col >> 5] >> (col & 31u)) & 1u) s_hits[atomicAdd(s_nhit, 1u)] = q;
        }
        __syncthreads();
        const uint32_t nhit = *s_nhit;
        __syncthreads();
        if (tid == 0) *s_nhit = 0u;
        for (uint32_t h = 0; h < nhit; h++) {
            const uint4 ch = a.chunks[s_hits[h]];
            for (uint32_t k = tid; k < ch.z; k += T) {
                const uint2 r0 = load_stream_nt(a.stream + ch.y + k);
                bfs_claim(a, bfs_candidate(a, true, r0), r0.x, fresh, work, work_rows);
            }
        }
        __syncthreads();
    }
}

// The bottom-up scan of the same kernel: a thread per row not reached yet, 64 rows per wavefront = one 64-bit word of the
// next frontier (shard bounds are multiples of 64 rows: every word has one writer).  Wavefront `wv` of `nwv`; GL_BFS_BU_ROWS words
// per step, every load of a stage issued for all of them before the first use (few wavefronts fit next to the pull's LDS tile).
#ifndef GL_BFS_BU_ROWS
#define GL_BFS_BU_ROWS 2
#endif
__device__ __forceinline__ void bfs_shard_bottom_up(const BfsPushArgs &a, uint32_t wv, uint32_t nwv, uint32_t &fresh, uint32_t &work,
                                                    uint32_t &work_rows) {
    const uint32_t lane = threadIdx.x & 63u;
    constexpr int R = GL_BFS_BU_ROWS;   // words (rows per lane) per step
    const uint32_t nwords64 = (a.row_end + 63u) >> 6;
    for (uint32_t wd0 = (a.row_begin >> 6) + wv; wd0 < nwords64; wd0 += (uint32_t)R * nwv) {
        uint32_t row[R], beg[R], end[R], len[R];
        bool hit[R], live[R];
        float dv[R];
#pragma unroll
        for (int u = 0; u < R; u++) {
            row[u] = (wd0 + u * nwv) * 64u + lane;
            live[u] = row[u] < a.row_end;
            dv[u] = live[u] ? a.dist[row[u]] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
            live[u] = live[u] && dv[u] == 0.0f;
            beg[u] = live[u] ? a.row_ptr[row[u] - a.row_begin] - a.nz_base : 0u;
            end[u] = live[u] ? a.row_ptr[row[u] - a.row_begin + 1u] - a.nz_base : 0u;
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
            len[u] = end[u] - beg[u];
            hit[u] = false;
        }
        auto undecided = [&]() {
            bool any = false;
#pragma unroll
            for (int u = 0; u < R; u++) any = any || (!hit[u] && beg[u] < end[u]);
            return any;
        };
        for (int step = 0; step < 8 && __any(undecided()); step++) {
            uint32_t c[R][4];
#pragma unroll
            for (int u = 0; u < R; u++)
#pragma unroll
                for (int k = 0; k < 4; k++) c[u][k] = (!hit[u] && beg[u] + k < end[u]) ? a.row_idx[beg[u] + k] : 0xffffffffu;
            uint32_t any[R];
#pragma unroll
            for (int u = 0; u < R; u++) any[u] = 0u;
#pragma unroll
            for (int u = 0; u < R; u++)
#pragma unroll
                for (int k = 0; k < 4; k++) any[u] |= c[u][k] < a.num_cols ? (a.bits_in[c[u][k] >> 5] >> (c[u][k] & 31u)) & 1u : 0u;
#pragma unroll
            for (int u = 0; u < R; u++) {
                if (!hit[u] && beg[u] < end[u]) {
                    hit[u] = any[u] != 0u;
                    beg[u] += 4u;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
            const uint32_t wd = wd0 + u * nwv;
            if (wd >= nwords64) continue;   // (wave-uniform)
            // rows still undecided after 32 entries are finished by the whole wavefront, 256 entries per step
            for (uint64_t pending = __ballot(!hit[u] && beg[u] < end[u]); pending; pending &= pending - 1ull) {
                const int src = __ffsll((unsigned long long)pending) - 1;
                const uint32_t b = __shfl(beg[u], src), e = __shfl(end[u], src);
                bool found = false;
                for (uint32_t base = b; base < e && !found; base += 256u) {
                    uint32_t any = 0u;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t i = base + 64u * k + lane;
                        const uint32_t c = i < e ? a.row_idx[i] : 0xffffffffu;
                        any |= c < a.num_cols ? (a.bits_in[c >> 5] >> (c & 31u)) & 1u : 0u;
                    }
                    found = __any(any != 0u);
                }
                if ((int)lane == src) hit[u] = found;
            }
            if (hit[u]) {
                a.dist[row[u]] = a.level;
                if (!a.deferred) {
                    fresh += 1u;
                    work_rows += len[u];
                    if (a.col_len) work += a.col_len[row[u]];
                    else if (row[u] < a.num_cols) work += a.indptr[row[u] + 1u] - a.indptr[row[u]];
                }
            }
            const uint64_t m = __ballot(hit[u]);
            if (lane == 0) reinterpret_cast<uint64_t *>(a.bits_out)[wd] = m;
        }
    }
}

}  // namespace gl

#endif  // GL_BFS_SHARD_H_
