// SpMSpV as propagation blocking (round 4): BIN the products by row tile, then FOLD every tile in LDS.
//
// Replaces the scatter of rounds 1-3 (one global atomic per product into a dense accumulator + an n-wide count / scan /
// write compaction).  hw/kernel_spmspv_impl.h:448-562 does the same two things per row tile of its output buffer: the PEs
// accumulate a tile (:492-540), then checkout_results + write_back_gmem compact it behind the tiles before it (:150-289).
//
//   bin   (spmspv_bin_kernel)   workgroups of 1024 threads take slices of the sparse input vector, read the active columns'
//         {row, value} runs coalesced (a lane per product, the lane's column found by a binary search in the slice's prefix
//         of column lengths), form the products, and sort a batch of 8192 of them by ROW TILE in LDS (counting sort: LDS
//         histogram, scan, one global reservation per tile that occurs, sorted staging buffer) -- the batch then leaves as
//         one contiguous run per tile into that tile's BIN.  A bin holds as many records as its row tile holds non-zeros, so
//         it cannot overflow unless the vector names a column twice; products that find no room go to the dense
//         accumulator with global atomics (slow, never wrong).  Columns of 2048 entries and more are cut into chunks that
//         all workgroups drain from a queue once every workgroup has queued its own.
//   fold  (spmspv_fold_kernel)  one workgroup per row tile: the tile's accumulators live in LDS (Tile<OP>: f64 adds for
//         (+,x), ordered-integer min, stores), the bin is streamed into them, rows with a result != zero that the mask
//         (compared with `zero`, spmspv_module.h:499-516) allows are counted, the count is published, the workgroup adds up
//         the counts of the tiles before it (they run at the same time) and writes its {index, value} run there: the
//         result list is in ascending row order with no n-wide pass.  The same kernel merges what a row-wise run (the
//         operator's direction switch) or an overflow left in the dense accumulator, does the fused sparse assign, leaves
//         the emitted rows as bits, takes the driver's loop decision (gl_compact.h Direction), and -- last workgroup to
//         finish -- restores every "zero between runs" word and, for a blocking caller, stores {sequence, count} to
//         page-locked host memory so that the host need not wait for the stream (profiles/r03_ubench_sync.txt).
//
// Bytes per product: 8 read from the column stream + 8 written to a bin + 8 read back = 24, all in runs.
#ifndef GL_SPMSPV_BIN_H_
#define GL_SPMSPV_BIN_H_

#include "gl_common.h"
#include "gl_compact.h"
#include "gl_tile.h"

namespace gl {

constexpr uint32_t kBinThreads = 1024;
constexpr uint32_t kBinItems = 8;                          // products per thread and batch
constexpr uint32_t kBinBatch = kBinThreads * kBinItems;    // 8192 products are sorted by tile at a time
constexpr uint32_t kBinMaxTiles = 2048;                    // LDS counters of the bin kernel
constexpr uint32_t kBinSlice = 1024;                       // vector entries a workgroup stages at a time
constexpr uint32_t kBigColumn = 2048;                      // columns at least this long go to the chunk queue
constexpr uint32_t kChunk = kBinBatch;                     // entries per queued chunk = one batch
constexpr uint32_t kFoldThreads = 1024;
constexpr uint32_t kFoldMaxRows = 16384;                   // rows per tile: 128 KB of 8-byte accumulators
constexpr uint32_t kFoldWaves = kFoldThreads / 64;
constexpr uint32_t kFoldMaxRounds = kFoldMaxRows / kFoldThreads;   // 16 rounds of 1024 rows

// words of gl_spmspv_plan_s::d_sync, all zero between runs
enum : uint32_t { kSyncQueued = 0, kSyncQueueHead = 1, kSyncProducers = 2, kSyncFoldDone = 3, kSyncFoldTicket = 4, kSyncTotal = 5, kSyncWords = 16 };

// row -> tile without a division: tile = (row * magic) >> (32 + shift), checked on the host for every tile boundary at plan
// creation (the function is monotone, so exact boundaries make it exact everywhere); magic == 0: rows per tile is 1 << shift
struct TileMap {
    uint32_t rows = 64, magic = 0, shift = 6, count = 1;
    __host__ __device__ uint32_t of(uint32_t r) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return (magic ? __umulhi(r, magic) : r) >> shift;
#else
        return magic ? (uint32_t)(((unsigned long long)r * magic) >> 32) >> shift : r >> shift;
#endif
    }
};

struct BinArgs {
    const uint32_t *indptr;
    const uint2 *stream;
    const gl_idx_val *vec;
    uint2 *bins;                // one record {shard-local row, product} per product
    const uint32_t *bin_base;   // tiles + 1 offsets into bins (capacity of tile t = bin_base[t + 1] - bin_base[t])
    uint32_t *cursor;           // records reserved per tile (may exceed the capacity: the surplus went to acc)
    float *acc;                 // dense accumulator of the shard's rows
    uint32_t *sync;
    unsigned long long *queue;  // two words per chunk: {first entry | count << 32, value bits}
    uint32_t queue_capacity;    // 0: the vector is known to name no long column (no queue phase)
    TileMap tiles;
    uint32_t binned;            // 0: more tiles than the kernel has counters for -- every product goes to acc
    uint32_t row_begin, num_cols;
    const uint32_t *mode;       // non-null: skip when mode[0] != 0 (the run goes row-wise instead)
    Gate gate;
};

// ordered-integer trick: for IEEE floats, a >= 0 compares like int, a < 0 like reversed uint
__device__ __forceinline__ void atomic_min_float(float *addr, float v) {
    if (!(__float_as_uint(v) >> 31))   // by sign bit, so that -0.0 takes the negative path (v >= 0 is true for it)
        atomicMin((int *)addr, __float_as_int(v));
    else
        atomicMax((unsigned int *)addr, __float_as_uint(v));
}

// a (x) x as the SpMSpV PE forms it; false: the product cannot change any result (a false && b)
template <int OP>
__device__ __forceinline__ bool spmspv_product(float a, float xv, float &z) {
    if (OP == GL_OP_MULADD) {
        z = a * xv;
        return true;
    } else if (OP == GL_OP_ANDOR) {
        z = 1.0f;
        return a != 0.0f && xv != 0.0f;
    } else if (OP == GL_OP_ADDMIN) {
        // saturating add of the (min,+) PE: hw/float_pe.h:24-33, spmspv_module.h:482-491
        if (a > kFloatInf || xv > kFloatInf) {
            z = kFloatInf;
        } else {
            z = a + xv;
            if (z > kFloatInf) z = kFloatInf;
        }
        return true;
    } else if (OP == kOpU32AndOr || OP == kOpFixAndOr) {
        z = bitsf(OP == kOpU32AndOr ? 1u : kFixOne);
        return fbits(a) != 0u && fbits(xv) != 0u;
    } else {
        z = Semiring<OP>::mul(a, xv);
        return OP != kOpFixMulAdd || fbits(z) != 0u;
    }
}

// acc[row] (+)= z with global atomics (z = a product already formed)
template <int OP>
__device__ __forceinline__ void spill_one(float *acc, uint32_t row, float z) {
    if (OP == kOpU32MulAdd) {
        atomicAdd(reinterpret_cast<unsigned int *>(&acc[row]), fbits(z));
    } else if (OP == kOpFixMulAdd) {
        // acc = min(acc + z, 2^32 - 1): clamped adds of non-negative terms give min(sum, 2^32 - 1) in any order (gl_common.h)
        unsigned int *w = reinterpret_cast<unsigned int *>(&acc[row]);
        unsigned int old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (old != 0xffffffffu) {
            const unsigned int seen = atomicCAS(w, old, sat_add_u32(old, fbits(z)));
            if (seen == old) break;
            old = seen;
        }
    } else if (OP == kOpU32AndOr || OP == kOpFixAndOr || OP == GL_OP_ANDOR) {
        acc[row] = z;
    } else if (OP == kOpU32AddMin || OP == kOpFixAddMin) {
        atomicMin(reinterpret_cast<unsigned int *>(&acc[row]), fbits(z));
    } else if (OP == GL_OP_MULADD) {
        unsafeAtomicAdd(&acc[row], z);
    } else {
        atomic_min_float(&acc[row], z);
    }
}

// exclusive prefix of v over the 1024 threads of the block, total in *total (s_wave: 16 words); two barriers
__device__ __forceinline__ uint32_t block_exclusive_1024(uint32_t v, uint32_t *s_wave, uint32_t *total) {
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
    for (uint32_t k = 0; k < 16; k++) {
        const uint32_t c = s_wave[k];
        if (k < w) before += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return before + incl - v;
}

struct BinLds {
    uint32_t start[kBinSlice];       // first stream entry of the slice's columns
    uint32_t pref[kBinSlice + 1];    // exclusive prefix of their lengths (long columns count 0)
    float val[kBinSlice];            // their vector values
    uint32_t cnt[kBinMaxTiles];      // records of the batch per tile; zero between batches
    uint32_t lbase[kBinMaxTiles];    // where the tile's records start in `sorted`
    uint32_t dest[kBinMaxTiles];     // bins index of sorted[i] = dest[tile] + i (mod 2^32)
    uint32_t lim[kBinMaxTiles];      // first bins index past the tile's bin
    uint32_t aux[kBinSlice];         // (queue overflow only: the columns' queue slots)
    uint2 sorted[kBinBatch];
    uint32_t wave[16];
    uint32_t word;
};

// One batch: the thread's kBinItems products {row (shard-local), z} with their `ok` flags -> bins.
template <int OP>
__device__ __forceinline__ void bin_batch(const BinArgs &a, BinLds &L, const uint32_t (&row)[kBinItems], const float (&z)[kBinItems],
                                          const bool (&ok)[kBinItems]) {
    const uint32_t tid = threadIdx.x;
    if (!a.binned) {
#pragma unroll
        for (uint32_t k = 0; k < kBinItems; k++)
            if (ok[k]) spill_one<OP>(a.acc, row[k], z[k]);
        return;
    }
    uint32_t t[kBinItems], rank[kBinItems];
#pragma unroll
    for (uint32_t k = 0; k < kBinItems; k++) {
        t[k] = a.tiles.of(row[k]);
        rank[k] = ok[k] ? atomicAdd(&L.cnt[t[k]], 1u) : 0u;
    }
    __syncthreads();
    // tiles 2 tid and 2 tid + 1: counts -> one reservation each in the tiles' bins (issued before the scan's barriers)
    uint32_t c0 = 0, c1 = 0, g0 = 0, g1 = 0, b0 = 0, b1 = 0, b2 = 0;
    const uint32_t t0 = 2u * tid;
    if (t0 < a.tiles.count) {
        c0 = L.cnt[t0];
        b0 = a.bin_base[t0];
        b1 = a.bin_base[t0 + 1u];
        if (c0) {
            L.cnt[t0] = 0u;
            g0 = atomicAdd(&a.cursor[t0], c0);
        }
        if (t0 + 1u < a.tiles.count) {
            c1 = L.cnt[t0 + 1u];
            b2 = a.bin_base[t0 + 2u];
            if (c1) {
                L.cnt[t0 + 1u] = 0u;
                g1 = atomicAdd(&a.cursor[t0 + 1u], c1);
            }
        }
    }
    uint32_t total;
    const uint32_t before = block_exclusive_1024(c0 + c1, L.wave, &total);
    if (t0 < a.tiles.count) {
        L.lbase[t0] = before;
        L.dest[t0] = b0 + g0 - before;
        L.lim[t0] = b1;
        if (t0 + 1u < a.tiles.count) {
            L.lbase[t0 + 1u] = before + c0;
            L.dest[t0 + 1u] = b1 + g1 - (before + c0);
            L.lim[t0 + 1u] = b2;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kBinItems; k++)
        if (ok[k]) L.sorted[L.lbase[t[k]] + rank[k]] = make_uint2(row[k], fbits(z[k]));
    __syncthreads();
    // consecutive lanes write consecutive records: one run per tile
#pragma unroll
    for (uint32_t k = 0; k < kBinItems; k++) {
        const uint32_t i = k * kBinThreads + tid;
        if (i < total) {
            const uint2 e = L.sorted[i];
            const uint32_t tt = a.tiles.of(e.x);
            const uint32_t pos = L.dest[tt] + i;
            if (pos < L.lim[tt])
                a.bins[pos] = e;
            else
                spill_one<OP>(a.acc, e.x, bitsf(e.y));   // the bin is full (a column named twice): dense accumulator
        }
    }
    // (no barrier here: the next batch passes two barriers before it writes lbase / dest / lim and three before `sorted`)
}

// a queued chunk (or a long column's chunk that found no queue slot): `count` <= kChunk consecutive stream entries times xv
template <int OP>
__device__ __forceinline__ void bin_chunk(const BinArgs &a, BinLds &L, uint32_t first, uint32_t count, float xv) {
    uint32_t row[kBinItems];
    float z[kBinItems];
    bool ok[kBinItems];
    uint2 rv[kBinItems];
#pragma unroll
    for (uint32_t k = 0; k < kBinItems; k++) {
        const uint32_t i = k * kBinThreads + threadIdx.x;
        ok[k] = i < count;
        rv[k] = ok[k] ? load_stream_nt(a.stream + first + i) : make_uint2(a.row_begin, 0u);
    }
#pragma unroll
    for (uint32_t k = 0; k < kBinItems; k++) {
        row[k] = rv[k].x - a.row_begin;
        ok[k] = ok[k] && spmspv_product<OP>(bitsf(rv[k].y), xv, z[k]);
    }
    bin_batch<OP>(a, L, row, z, ok);
}

template <int OP>
__global__ __launch_bounds__(kBinThreads) void spmspv_bin_kernel(BinArgs a) {
    __shared__ BinLds L;
    if (a.gate.closed()) return;
    if (a.mode && a.mode[0]) return;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < kBinMaxTiles; i += kBinThreads) L.cnt[i] = 0u;
    const uint32_t vnnz = a.vec[0].index;
    // a small vector is spread over the whole grid (E entries per workgroup, down to one)
    const uint32_t E = min(kBinSlice, max(1u, (vnnz + gridDim.x - 1u) / gridDim.x));
    uint32_t steps = 0;          // of the binary search: the slice holds E <= 2^steps columns
    while ((1u << steps) < E) steps++;
    __syncthreads();
    for (uint32_t slice = blockIdx.x * E; slice < vnnz; slice += gridDim.x * E) {
        uint32_t start = 0, deg = 0;
        float xv = 0.0f;
        if (tid < E && slice + tid < vnnz) {
            const gl_idx_val iv = a.vec[1u + slice + tid];
            if (iv.index < a.num_cols) {
                start = a.indptr[iv.index];
                deg = a.indptr[iv.index + 1u] - start;
                xv = iv.val;
            }
        }
        const bool big = a.queue_capacity != 0u && deg >= kBigColumn;
        if (a.queue_capacity) {
            // long columns -> queue chunks (one atomic per workgroup reserves the slots)
            const uint32_t nchunks = big ? (deg + kChunk - 1u) / kChunk : 0u;
            uint32_t qtotal;
            const uint32_t qoff = block_exclusive_1024(nchunks, L.wave, &qtotal);
            if (qtotal) {   // block-uniform
                if (tid == 0) L.word = atomicAdd(&a.sync[kSyncQueued], qtotal);
                __syncthreads();
                const uint32_t qb = L.word + qoff;
                for (uint32_t c = 0; c < nchunks; c++) {
                    if (qb + c < a.queue_capacity) {
                        const unsigned long long d0 = (unsigned long long)(start + c * kChunk) |
                                                      ((unsigned long long)min(kChunk, deg - c * kChunk) << 32);
                        // (agent-scope stores: another compute unit reads them after the producers' counter, no fence)
                        __hip_atomic_store(&a.queue[2ull * (qb + c)], d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&a.queue[2ull * (qb + c) + 1ull], (unsigned long long)fbits(xv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                // The queue holds one slot per chunk of every long column (exact, computed at plan creation), so it can only
                // run out when the input vector names a column more than once.  Chunks without a slot are binned right
                // here by this workgroup.
                if (L.word + qtotal > a.queue_capacity) {   // block-uniform
                    L.pref[tid] = nchunks ? deg : 0u;
                    L.start[tid] = start;
                    L.val[tid] = xv;
                    L.aux[tid] = qb;
                    __syncthreads();
                    for (uint32_t j = 0; j < kBinSlice; j++) {
                        const uint32_t dj = L.pref[j];
                        if (!dj) continue;
                        const uint32_t nj = (dj + kChunk - 1u) / kChunk, qj = L.aux[j];
                        const uint32_t c0 = qj >= a.queue_capacity ? 0u : min(nj, a.queue_capacity - qj);   // first chunk without a slot
                        const uint32_t sj = L.start[j];
                        const float xj = L.val[j];
                        for (uint32_t c = c0; c < nj; c++) {
                            __syncthreads();
                            bin_chunk<OP>(a, L, sj + c * kChunk, min(kChunk, dj - c * kChunk), xj);
                        }
                    }
                }
                __syncthreads();
            }
        }
        // the rest: a lane per product
        uint32_t W;
        const uint32_t pre = block_exclusive_1024(big ? 0u : deg, L.wave, &W);
        L.start[tid] = start;
        L.pref[tid] = pre;
        L.val[tid] = xv;
        if (tid == kBinThreads - 1u) L.pref[kBinSlice] = W;
        __syncthreads();
        for (uint32_t w0 = 0; w0 < W; w0 += kBinBatch) {
            uint32_t row[kBinItems];
            float z[kBinItems], xs[kBinItems];
            bool ok[kBinItems];
            uint2 rv[kBinItems];
#pragma unroll
            for (uint32_t k = 0; k < kBinItems; k++) {
                const uint32_t item = w0 + k * kBinThreads + tid;
                ok[k] = item < W;
                uint32_t lo = 0, hi = E - 1u;     // largest j with pref[j] <= item (zero-length columns share a prefix:
                for (uint32_t it = 0; it < steps; it++) {   // the largest such j owns the item)
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    if (L.pref[mid] <= item) lo = mid; else hi = mid - 1u;
                }
                xs[k] = L.val[lo];
                rv[k] = ok[k] ? load_stream_nt(a.stream + L.start[lo] + (item - L.pref[lo])) : make_uint2(a.row_begin, 0u);
            }
#pragma unroll
            for (uint32_t k = 0; k < kBinItems; k++) {
                row[k] = rv[k].x - a.row_begin;
                ok[k] = ok[k] && spmspv_product<OP>(bitsf(rv[k].y), xs[k], z[k]);
            }
            bin_batch<OP>(a, L, row, z, ok);
        }
        __syncthreads();
    }
    if (!a.queue_capacity) return;
    // ---- the chunk queue: every workgroup has queued its chunks once it arrives here; when all have, all drain
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's descriptor stores have been performed
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(&a.sync[kSyncProducers], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the grid is at most one workgroup per compute unit: all of them are resident)
        uint32_t spins = 0;
        while (__hip_atomic_load(&a.sync[kSyncProducers], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1u << 24))
            __builtin_amdgcn_s_sleep(2);
        L.word = min(__hip_atomic_load(&a.sync[kSyncQueued], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a.queue_capacity);
    }
    __syncthreads();
    const uint32_t nq = L.word;
    if (!nq) return;
    for (;;) {
        __syncthreads();
        if (tid == 0) L.word = atomicAdd(&a.sync[kSyncQueueHead], 1u);
        __syncthreads();
        const uint32_t q = L.word;
        if (q >= nq) break;
        const unsigned long long d0 = __hip_atomic_load(&a.queue[2ull * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long d1 = __hip_atomic_load(&a.queue[2ull * q + 1ull], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bin_chunk<OP>(a, L, (uint32_t)d0, (uint32_t)(d0 >> 32), bitsf((uint32_t)d1));
    }
}

// ------------------------------------------------------------------------------------------------------------------ fold
struct FoldArgs {
    const uint2 *bins;
    const uint32_t *bin_base;
    uint32_t *cursor;
    float *acc;               // dense accumulator (all == zero between runs): spilled products, or a row-wise run's result
    const float *mask;
    int mask_type;
    uint32_t nrows, row_begin;
    float zero;
    TileMap tiles;
    gl_idx_val *out;
    float head_val;
    float *assign;            // gl_spmspv_run_assign: assign[index] = assign_val for every emitted entry (or null)
    float assign_val;
    uint32_t *next_bits;      // the emitted rows also as a bit vector: every word of the shard's rows is written (or null)
    uint32_t *state;          // tiles words: 1 << 31 | entries of the tile once it has counted; zero between runs
    uint32_t *sync;
    uint32_t tickets;         // more tiles than resident workgroups: tiles are handed out in arrival order
    uint32_t merge_all;       // every tile also takes what the dense accumulator holds
    const uint32_t *mode;     // non-null: the same when mode[0] != 0 (the run went row-wise)
    unsigned long long *host_rec;   // page-locked host word: seq << 32 | count when everything has been written (or null)
    uint32_t seq;
    Gate gate;
    Direction dir;
};

template <int OPX>
__global__ __launch_bounds__(kFoldThreads) void spmspv_fold_kernel(FoldArgs a) {
    using TL = Tile<OPX>;
    using T = typename TL::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char fold_lds_raw[];
    T *tile = reinterpret_cast<T *>(fold_lds_raw);
    __shared__ unsigned long long s_ball[kFoldMaxRounds * kFoldWaves];   // keep-ballot of (round, wavefront)
    __shared__ uint32_t s_off[kFoldMaxRounds * kFoldWaves];              // entries in front of it inside the tile
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_word;
    if (a.gate.closed()) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr bool BITS = OPX >= 3;   // the integer value types compare bit patterns (zero may be a NaN as a float)
    uint32_t t = blockIdx.x;
    if (a.tickets) {
        if (tid == 0) s_word = atomicAdd(&a.sync[kSyncFoldTicket], 1u);
        __syncthreads();
        t = s_word;
    }
    const uint32_t T_ = a.tiles.count, R = a.tiles.rows;
    const uint32_t row0 = t * R;
    const uint32_t rows = a.nrows > row0 ? min(R, a.nrows - row0) : 0u;
    const uint32_t rounds = (rows + kFoldThreads - 1u) / kFoldThreads;
    const uint32_t raw = a.cursor[t];
    const uint32_t bb = a.bin_base[t], cap = a.bin_base[t + 1u] - bb;
    const uint32_t cnt = min(raw, cap);
    const bool merge = a.merge_all != 0u || (a.mode && a.mode[0]) || raw > cap;
    uint32_t total = 0;
    if (cnt || merge) {
        for (uint32_t i = tid; i < rows; i += kFoldThreads) tile[i] = TL::ident();
        __syncthreads();
        if (tid == 0 && raw) a.cursor[t] = 0u;   // (behind the barrier: every wavefront has read it)
        // ---- the bin -> the tile's accumulators (4 records per thread in flight)
        const uint2 *bin = a.bins + bb;
        for (uint32_t i0 = 0; i0 < cnt; i0 += 4u * kFoldThreads) {
            uint2 e[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const uint32_t i = i0 + u * kFoldThreads + tid;
                e[u] = i < cnt ? load_stream_nt(bin + i) : make_uint2(0xffffffffu, 0u);
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++)
                if (e[u].x != 0xffffffffu) TL::accz(tile, e[u].x - row0, bitsf(e[u].y));
        }
        __syncthreads();
        // ---- per row: the value, the mask (compared with `zero`, hw/kernel_spmspv_impl.h:262-283), the keep flag; the value
        // goes back into the row's LDS slot as a float for the write pass
        for (uint32_t j = 0; j < rounds; j++) {
            const uint32_t r = j * kFoldThreads + tid;
            bool keep = false;
            if (r < rows) {
                const float s = TL::get(tile, r);
                float v;
                if (merge) {
                    const float w = a.acc[row0 + r];
                    if (fbits(w) != fbits(a.zero)) a.acc[row0 + r] = a.zero;
                    v = Semiring<OPX>::add(w, s);
                } else {
                    v = TL::finish(a.zero, s);
                }
                keep = BITS ? (fbits(v) != fbits(a.zero)) : (v != a.zero);   // checkout_results: dense_data != zero (:199-226)
                if (keep && a.mask_type != GL_NOMASK) {
                    const float m = a.mask[a.row_begin + row0 + r];
                    const bool eq = BITS ? (fbits(m) == fbits(a.zero)) : (m == a.zero);
                    keep = a.mask_type == GL_MASK_WRITETOZERO ? eq : !eq;
                }
                *reinterpret_cast<float *>(&tile[r]) = v;
            }
            const unsigned long long b = __ballot(keep);
            if (lane == 0) s_ball[j * kFoldWaves + wave] = b;
        }
        __syncthreads();
        // exclusive prefix of the (round, wavefront) counts: at most 256 of them
        {
            const uint32_t c = tid < rounds * kFoldWaves ? (uint32_t)__popcll(s_ball[tid]) : 0u;
            const uint32_t before = block_exclusive_1024(c, s_wave, &total);
            if (tid < rounds * kFoldWaves) s_off[tid] = before;
        }
    }
    // ---- publish the tile's count; entries of the tiles in front = this tile's place in the list
    if (tid == 0) __hip_atomic_store(&a.state[t], 0x80000000u | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t before = 0;
    for (uint32_t u = tid; u < t; u += kFoldThreads) {
        uint32_t w, spins = 0;
        while (!((w = __hip_atomic_load(&a.state[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 31) && ++spins < (1u << 24))
            __builtin_amdgcn_s_sleep(1);
        before += w & 0x7fffffffu;
    }
    if (t > 0u) {   // block-uniform
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) before += __shfl_down(before, d);
        __syncthreads();
        if (lane == 0) s_wave[wave] = before;
        __syncthreads();
        before = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) before += s_wave[k];
    }
    // ---- the write pass
    const uint32_t row_g0 = a.row_begin + row0;
    if (total) {
        for (uint32_t j = 0; j < rounds; j++) {
            const uint32_t r = j * kFoldThreads + tid;
            const unsigned long long b = s_ball[j * kFoldWaves + wave];
            if ((b >> lane) & 1ull) {
                gl_idx_val item;
                item.index = row_g0 + r;
                item.val = *reinterpret_cast<const float *>(&tile[r]);
                a.out[1u + before + s_off[j * kFoldWaves + wave] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = item;
                if (a.assign) a.assign[item.index] = a.assign_val;   // the entry's own row: nobody else touches it
            }
        }
    }
    if (a.next_bits) {
        // every 32-bit word of the shard's rows is (re)written: R is a multiple of 64, row_begin of 32
        const uint32_t groups = (min(R, a.nrows > row0 ? a.nrows - row0 : 0u) + 63u) / 64u;
        for (uint32_t g = tid; g < 2u * groups; g += kFoldThreads) {
            const uint32_t gi = g >> 1, half = g & 1u;
            if (row0 + gi * 64u + half * 32u >= a.nrows) continue;
            const unsigned long long b = total ? s_ball[(gi >> 4) * kFoldWaves + (gi & 15u)] : 0ull;
            a.next_bits[(row_g0 >> 5) + 2u * gi + half] = (uint32_t)(b >> (32u * half));
        }
    }
    if (t == T_ - 1u && tid == 0) {
        a.out[0].index = before + total;
        a.out[0].val = a.head_val;
        a.dir.decide(before + total);
        __hip_atomic_store(&a.sync[kSyncTotal], before + total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the last workgroup to finish restores the "zero between runs" words and tells a blocking caller
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's stores have been performed
    __syncthreads();
    if (tid == 0) s_word = __hip_atomic_fetch_add(&a.sync[kSyncFoldDone], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_word != T_ - 1u) return;
    for (uint32_t u = tid; u < T_; u += kFoldThreads) __hip_atomic_store(&a.state[u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        const uint32_t all = __hip_atomic_load(&a.sync[kSyncTotal], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t k = 0; k < 6u; k++) __hip_atomic_store(&a.sync[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.host_rec) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(a.host_rec, ((unsigned long long)a.seq << 32) | all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace gl

#endif  // GL_SPMSPV_BIN_H_
