// SpMSpV as propagation blocking (round 4): BIN the products by row tile, then FOLD every tile in LDS.
//
// Replaces the scatter of rounds 1-3 (one global atomic per product into a dense accumulator + an n-wide count / scan /
// write compaction).  hw/kernel_spmspv_impl.h:448-562 does the same two things per row tile of its output buffer: the PEs
// accumulate a tile (:492-540), then checkout_results + write_back_gmem compact it behind the tiles before it (:150-289).
//
//   bin   (spmspv_bin_kernel)   the run's products, numbered in vector order, are cut into equal ranges, one per workgroup of
//         1024 threads (one rendezvous of the grid gives every workgroup the prefix of the column lengths over the vector: a
//         hub column is shared by as many workgroups as its length asks for).  A workgroup reads its range of the active
//         columns' {row, value} runs coalesced (a lane per product, the lane's column found by a binary search in the staged
//         prefix), forms the products, and sorts a batch of 8192 of them by ROW TILE in LDS (counting sort: LDS histogram,
//         scan, one global reservation per tile that occurs, sorted staging buffer) -- the batch then leaves as one
//         contiguous run per tile into that tile's BIN.  A bin holds as many records as its row tile holds non-zeros, so it
//         cannot overflow unless the vector names a column twice; products that find no room go to the dense accumulator
//         with global atomics (slow, never wrong).
//   fold  (spmspv_fold_kernel)  one workgroup per row tile: the tile's accumulators live in LDS (Tile<OP>: f64 adds for
//         (+,x), ordered-integer min, stores), the bin is streamed into them, rows with a result != zero that the mask
//         (compared with `zero`, spmspv_module.h:499-516) allows are counted, the count is published, the workgroup adds up
//         the counts of the tiles before it (they run at the same time) and writes its {index, value} run there: the
//         result list is in ascending row order with no n-wide pass.  The same kernel merges what a row-wise run (the
//         operator's direction switch) or an overflow left in the dense accumulator, does the fused sparse assign, and --
//         last workgroup to finish, for a blocking caller -- stores {sequence, count} to page-locked host memory so that
//         the host need not wait for the stream (profiles/r03_ubench_sync.txt).
//
// Bytes per product: 8 read from the column stream + 8 written to a bin + 8 read back = 24, all in runs.
#ifndef GL_SPMSPV_BIN_H_
#define GL_SPMSPV_BIN_H_

#include "gl_common.h"
#include "gl_tile.h"

namespace gl {

// -DGL_STAMPS (scratch builds, scripts/spmspv_stamps.py): thread 0 of every workgroup leaves wall_clock64() (100 MHz) at the phase
// boundaries of the bin (k = 0) and fold (k = 1) kernels; gl_debug_stamps copies them out
#if defined(GL_STAMPS)
__device__ unsigned long long g_stamps[2][256][16];
#define GL_STAMP(k, p)                                                                              \
    do {                                                                                            \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                 \
        if (threadIdx.x == 0 && blockIdx.x < 256u) g_stamps[k][blockIdx.x][p] = wall_clock64();     \
    } while (0)
#else
#define GL_STAMP(k, p) do { } while (0)
#endif
// the wavefront of a workgroup that polls the other workgroups' words (the publishing thread is in wavefront 0)
#if !defined(GL_POLL_WAVE)
#define GL_POLL_WAVE 0
#endif

constexpr uint32_t kBinThreads = 1024;
constexpr uint32_t kBinItems = 8;                          // products per thread and batch
constexpr uint32_t kBinBatch = kBinThreads * kBinItems;    // 8192 products are sorted by tile at a time
constexpr uint32_t kBinMaxTiles = 2048;                    // LDS counters of the bin kernel
constexpr uint32_t kBinSlice = 1024;                       // vector entries a workgroup stages at a time
constexpr uint32_t kBinLocalWindows = 4;                   // vectors of up to 4 x 1024 entries need no rendezvous (16: R5.11, slower)
constexpr uint32_t kBinByEntries = 128;                    // cut by entries: at most this many vector entries per workgroup ...
constexpr uint32_t kBinByEntriesMaxCol = 12288;            // ... and no column longer than a batch and a half (else: equal PRODUCT ranges)
constexpr uint32_t kBinMaxSlices = 2048;                   // slices of the vector per rendezvous (two per thread)
constexpr uint32_t kFoldThreads = 1024;
constexpr uint32_t kFoldMaxRows = 16384;                   // rows per tile: 128 KB of 8-byte accumulators
constexpr uint32_t kFoldWaves = kFoldThreads / 64;
constexpr uint32_t kFoldMaxRounds = kFoldMaxRows / kFoldThreads;   // 16 rounds of 1024 rows

// words of gl_spmspv_plan_s::d_sync.  kSyncGen: the generation that tags the bin kernel's slice sums and the fold kernel's tile
// states, starts at 1 and only grows (the fold's last tile advances it); kSyncFoldTicket: tiles handed out so far (64 bits,
// only grows); kSyncFoldDone: finished workgroups << 32 | their entries (64 bits), zero between runs
// kSyncErr: a rendezvous poll gave up (its spin limit ran out -- the all-workgroups-resident assumption was broken by a co-resident
// kernel): the run's list is emptied and a blocking caller gets GL_ERR_HIP from gl_spmspv_wait instead of a wrong result.  The
// word holds the GENERATION of the failed run (generations start at 1 and only grow), so it needs no clearing: a run is failed iff
// the word equals its own generation, and a stale mark cannot empty the lists of later runs or graph replays (round 5 cleared a
// boolean only on runs that report to the host: one timeout inside a recorded BFS slot would have emptied every later replay).
// kSyncFailedRuns: failed runs so far, never cleared -- what gl_spmspv_failed_runs reads after non-reporting runs (graph replays).
enum : uint32_t { kSyncGen = 0, kSyncFoldDone = 4 /* 64 bits */, kSyncFoldTicket = 8 /* 64 bits */, kSyncErr = 12, kSyncFailedRuns = 13, kSyncWords = 16 };
constexpr uint32_t kSpinLimit = 1u << 24;

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
    unsigned long long *slices; // 2 x kBinMaxSlices words (by round parity): tag << 32 | non-zeros in the columns of a slice of the vector
    TileMap tiles;
    uint32_t binned;            // 0: more tiles than the kernel has counters for -- every product goes to acc
    uint32_t row_begin, num_cols;
    uint32_t max_col_len;       // longest column of the shard
    const uint32_t *mode;       // non-null: skip when mode[0] != 0 (the run goes row-wise instead)
    uint32_t by_entries;        // no column of the shard is longer than kBinByEntriesMaxCol: mid-size vectors are cut by ENTRIES
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
    uint2 sorted[kBinBatch];
    unsigned long long wave64[16];
    unsigned long long pre0, r_P, r_lo, r_hi;   // the rendezvous' results, from the workgroup's first wavefront
    uint32_t wave[16];
    uint32_t tab[kBinBatch / 64u + 4u];   // column of every 64th product of the batch
    uint32_t word, r_over;
};

// batches of up to this many products skip the LDS staging: sorting pays once a tile's run is a few hundred bytes long
__host__ __device__ inline uint32_t bin_direct_limit(uint32_t tiles) { return min(kBinBatch - 1u, max(3072u, 24u * tiles)); }

// One batch: the thread's kBinItems products {row (shard-local), z} with their `ok` flags -> bins.
// `sorted` (block-uniform): a well-filled batch goes through the LDS staging buffer and leaves as one run per tile; a batch of
// a few thousand products (a latency-bound run: ~10 records per tile) is stored straight from the registers, each record at
// its tile's reservation + its rank in the tile -- two barriers instead of five, no scan.
template <int OP>
__device__ __forceinline__ void bin_batch(const BinArgs &a, BinLds &L, const uint32_t (&row)[kBinItems], const float (&z)[kBinItems],
                                          const bool (&ok)[kBinItems], bool sorted, uint32_t stamp_base = 0u) {
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
    GL_STAMP(0, 7u + stamp_base);      // histogram done
    // tiles 2 tid and 2 tid + 1: counts -> one reservation each in the tiles' bins (issued before the scan's barriers)
    uint32_t c0 = 0, c1 = 0, g0 = 0, g1 = 0, b0 = 0, b1 = 0, b2 = 0;
    const uint32_t t0 = 2u * tid;
    if (t0 < a.tiles.count) {
        c0 = L.cnt[t0];
        if (t0 + 1u < a.tiles.count) c1 = L.cnt[t0 + 1u];
        if (c0 | c1) {
            b0 = a.bin_base[t0];
            b1 = a.bin_base[t0 + 1u];
            if (t0 + 1u < a.tiles.count) b2 = a.bin_base[t0 + 2u];
        }
        if (c0) {
            L.cnt[t0] = 0u;
            g0 = atomicAdd(&a.cursor[t0], c0);
        }
        if (c1) {
            L.cnt[t0 + 1u] = 0u;
            g1 = atomicAdd(&a.cursor[t0 + 1u], c1);
        }
    }
    GL_STAMP(0, 8u + stamp_base);      // reservations back
    if (!sorted) {
        if (c0) {
            L.dest[t0] = b0 + g0;
            L.lim[t0] = b1;
        }
        if (c1) {
            L.dest[t0 + 1u] = b1 + g1;
            L.lim[t0 + 1u] = b2;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kBinItems; k++) {
            if (!ok[k]) continue;
            const uint32_t pos = L.dest[t[k]] + rank[k];
            if (pos < L.lim[t[k]])
                a.bins[pos] = make_uint2(row[k], fbits(z[k]));
            else
                spill_one<OP>(a.acc, row[k], z[k]);   // the bin is full (a column named twice): dense accumulator
        }
        return;   // (the next batch passes a barrier before it writes dest / lim again)
    }
    uint32_t total;
    const uint32_t before = block_exclusive_1024(c0 + c1, L.wave, &total);
    if (c0) {
        L.lbase[t0] = before;
        L.dest[t0] = b0 + g0 - before;
        L.lim[t0] = b1;
    }
    if (c1) {
        L.lbase[t0 + 1u] = before + c0;
        L.dest[t0 + 1u] = b1 + g1 - (before + c0);
        L.lim[t0 + 1u] = b2;
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
                spill_one<OP>(a.acc, e.x, bitsf(e.y));
        }
    }
    // (no barrier here: the next batch passes two barriers before it writes lbase / dest / lim and three before `sorted`)
}

// `count` <= kBinBatch consecutive stream entries of one column times xv (the overflow path's unit of work)
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
    bin_batch<OP>(a, L, row, z, ok, count > bin_direct_limit(a.tiles.count));
}

// Stage `nent` <= 1024 vector entries (first stream entry, exclusive prefix of the column lengths, value) and bin the products
// of the window that fall into [lo, hi) -- product numbers count from `cur` at the window's first entry.  grid != 0: the
// window IS the vector (at most 1024 entries): the workgroup cuts [0, W) into equal ranges itself and takes range blockIdx.x of
// `grid` -- no word has to travel between workgroups.  Returns the products of the window.
template <int OP>
__device__ __forceinline__ uint32_t bin_window(const BinArgs &a, BinLds &L, const gl_idx_val *vec, uint32_t nent, unsigned long long cur,
                                               unsigned long long lo, unsigned long long hi, uint32_t grid) {
    const uint32_t tid = threadIdx.x;
    uint32_t start = 0, deg = 0;
    float xv = 0.0f;
    if (tid < nent) {
        const gl_idx_val iv = vec[tid];
        if (iv.index < a.num_cols) {
            start = a.indptr[iv.index];
            deg = a.indptr[iv.index + 1u] - start;
            xv = iv.val;
        }
    }
    uint32_t W;
    const uint32_t pw = block_exclusive_1024(deg, L.wave, &W);
    L.start[tid] = start;
    L.pref[tid] = pw;
    L.val[tid] = xv;
    __syncthreads();
    GL_STAMP(0, 4);      // window staged
    if (grid) {
        // equal ranges of products, at least 2048 (a batch that small is mostly fixed cost), whole groups of 64
        const uint32_t Q = (max((W + grid - 1u) / grid, 2048u) + 63u) & ~63u;
        lo = (unsigned long long)blockIdx.x * Q;
        hi = min(lo + Q, (unsigned long long)W);
        if (lo >= W) return W;
    }
    uint32_t steps = 0;          // of the binary search: nent <= 2^steps
    while ((1u << steps) < nent) steps++;
    const uint32_t direct_limit = bin_direct_limit(a.tiles.count);
    // my products inside the window: [a0, a1) of its W
    const uint32_t a0 = lo > cur ? (uint32_t)(lo - cur) : 0u;
    const uint32_t a1 = (uint32_t)min((unsigned long long)W, hi - cur);
    for (uint32_t w0 = a0; w0 < a1; w0 += kBinBatch) {
        uint32_t row[kBinItems];
        float z[kBinItems], xs[kBinItems];
        bool ok[kBinItems];
        uint2 rv[kBinItems];
        // the column of every 64th product of the batch (129 ten-step searches by three wavefronts), so that the
        // per-product search below only looks between its group's two marks -- a lane-wide ten-step search per
        // product was 3-5 us of instruction issue per batch (16 wavefronts x 8 products x 10 steps x 6 instructions)
        if (tid <= kBinBatch / 64u) {
            const uint32_t item = min(w0 + 64u * tid, a1 - 1u);
            uint32_t l2 = 0, h2 = nent - 1u;     // largest j with pref[j] <= item (zero-length columns share a prefix:
            for (uint32_t it = 0; it < steps; it++) {   // the largest such j owns the item)
                const uint32_t mid = (l2 + h2 + 1u) >> 1;
                if (L.pref[mid] <= item) l2 = mid; else h2 = mid - 1u;
            }
            L.tab[tid] = l2;
        }
        __syncthreads();
        const uint32_t sb = w0 == a0 ? 0u : 6u;   // (stamps: the first batch in 5 .. 9, the last one in 11 .. 15)
        GL_STAMP(0, 5u + sb);  // owner table
        const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (uint32_t k = 0; k < kBinItems; k++) {
            const uint32_t item = w0 + k * kBinThreads + tid;
            ok[k] = item < a1;
            const uint32_t g = k * (kBinThreads / 64u) + wv;        // wave-uniform
            uint32_t l2 = L.tab[g], h2 = L.tab[g + 1u];
            const uint32_t span = __builtin_amdgcn_readfirstlane(h2 - l2);
            for (uint32_t rem = span; rem != 0u; rem >>= 1) {       // wave-uniform trip count
                const uint32_t mid = (l2 + h2 + 1u) >> 1;
                if (L.pref[mid] <= item) l2 = mid; else h2 = mid - 1u;
            }
            xs[k] = L.val[l2];
            rv[k] = ok[k] ? load_stream_nt(a.stream + L.start[l2] + (item - L.pref[l2])) : make_uint2(a.row_begin, 0u);
        }
#pragma unroll
        for (uint32_t k = 0; k < kBinItems; k++) {
            row[k] = rv[k].x - a.row_begin;
            ok[k] = ok[k] && spmspv_product<OP>(bitsf(rv[k].y), xs[k], z[k]);
        }
        GL_STAMP(0, 6u + sb);  // stream in
        bin_batch<OP>(a, L, row, z, ok, a1 - w0 > direct_limit, sb);
        GL_STAMP(0, 9u + sb);  // batch stored
    }
    return W;
}

// The products of a run are numbered 0 .. P - 1 in vector order and cut into EQUAL ranges, one per workgroup: a hub column is
// shared by as many workgroups as its length asks for, a thousand short ones are one workgroup's batch -- no chunk queue, no
// workgroup that runs four batches while the others wait.  P needs a prefix over the whole vector, so the kernel has ONE
// rendezvous: the vector is cut into S <= 2048 slices of E entries; workgroup b adds up the column lengths of slices b,
// b + G, ... and publishes each sum as a tagged word; every workgroup polls all S words (the grid is at most one workgroup
// per compute unit: all are resident), scans them, finds the slice its range starts in and from there stages 1024 vector
// entries at a time (their prefix of column lengths in LDS), a lane per product, the lane's column by binary search.
// Tags: sync[kSyncGen] + round; the last workgroup to leave advances the generation, so the words need no reset.  Vectors of
// more than 2048 x E entries take several rounds; the words are DOUBLE-BUFFERED by round parity: a workgroup that leaves round r
// early publishes round r + 1 into the other half while slower ones still poll round r, and nobody can be two rounds ahead
// (round r + 1's rendezvous needs everybody's r + 1 publication, which a workgroup makes after its last poll of round r).
template <int OP>
__global__ __launch_bounds__(kBinThreads) void spmspv_bin_kernel(BinArgs a) {
    __shared__ BinLds L;
    if (a.mode && a.mode[0]) return;
    const uint32_t tid = threadIdx.x, G = gridDim.x, blk = blockIdx.x;
    GL_STAMP(0, 0);
    for (uint32_t i = tid; i < kBinMaxTiles; i += kBinThreads) L.cnt[i] = 0u;
    const uint32_t vnnz = a.vec[0].index;
    const uint32_t E = min(kBinSlice, max(1u, (vnnz + G - 1u) / G));
    __syncthreads();
    // a vector of at most 1024 entries (whose columns cannot hold 2^32 non-zeros): every workgroup stages ALL of it and cuts
    // the products itself -- the rendezvous below (slice sums out, everybody's sums in: ~8 us of a 13 us launch) is not needed
    if (vnnz <= kBinSlice && (unsigned long long)vnnz * a.max_col_len <= 0xfffffffeull) {
        if (vnnz) (void)bin_window<OP>(a, L, a.vec + 1u, vnnz, 0ull, 0ull, 0ull, G);
        return;
    }
    // up to four windows of 1024 entries: still no rendezvous -- every workgroup adds up the column lengths of ALL entries itself
    // (four entries per thread, one round of loads), cuts the products, and stages only the windows its range touches
    if (vnnz <= kBinLocalWindows * kBinSlice && (unsigned long long)vnnz * a.max_col_len <= 0xfffffffeull) {
        uint32_t deg[kBinLocalWindows];
#pragma unroll
        for (uint32_t w = 0; w < kBinLocalWindows; w++) {
            const uint32_t e = w * kBinSlice + tid;
            uint32_t col = 0xffffffffu;
            if (e < vnnz) col = a.vec[1u + e].index;
            deg[w] = col < a.num_cols ? a.indptr[col + 1u] - a.indptr[col] : 0u;
        }
#pragma unroll
        for (uint32_t w = 0; w < kBinLocalWindows; w++) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) deg[w] += __shfl_down(deg[w], d);
            if ((tid & 63u) == 0u) L.tab[(tid >> 6) * kBinLocalWindows + w] = deg[w];     // (tab: 16 x 4 words here)
        }
        __syncthreads();
        uint32_t tot[kBinLocalWindows], P = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBinLocalWindows; w++) {
            tot[w] = 0u;
            for (uint32_t k = 0; k < kBinThreads / 64u; k++) tot[w] += L.tab[k * kBinLocalWindows + w];
            P += tot[w];
        }
        __syncthreads();
        const uint32_t Q = (max((P + G - 1u) / G, 2048u) + 63u) & ~63u;
        const unsigned long long lo = (unsigned long long)blk * Q, hi = min(lo + Q, (unsigned long long)P);
        if (lo >= P) return;
        unsigned long long cur = 0ull;
#pragma unroll
        for (uint32_t w = 0; w < kBinLocalWindows; w++) {
            if (w * kBinSlice < vnnz && cur + tot[w] > lo && cur < hi) {      // block-uniform
                (void)bin_window<OP>(a, L, a.vec + 1u + w * kBinSlice, min(kBinSlice, vnnz - w * kBinSlice), cur, lo, hi, 0u);
                __syncthreads();
            }
            cur += tot[w];
        }
        return;
    }
    // mid-size vectors over a matrix without long columns: equal slices of the VECTOR, one per workgroup -- no prefix over the whole
    // vector, hence no rendezvous (8 us of a 22 us launch, EXPERIMENTS R5.11); the price is one workgroup's luck with its columns,
    // at most kBinByEntriesMaxCol products more than its neighbours (a batch and a half: ~7 us), which is why long-columned
    // matrices and longer vectors (where a slice's sum varies by more than that) keep the equal product ranges below
    if (a.by_entries && vnnz <= kBinByEntries * G) {
        const uint32_t Es = (vnnz + G - 1u) / G, e0 = blk * Es;
        if (e0 < vnnz) (void)bin_window<OP>(a, L, a.vec + 1u + e0, min(Es, vnnz - e0), 0ull, 0ull, ~0ull, 0u);
        return;
    }
    const uint32_t gen0 = a.sync[kSyncGen];
    uint32_t rounds = 0;
    for (unsigned long long rbase = 0; rbase < vnnz; rbase += (unsigned long long)kBinMaxSlices * E, rounds++) {
        const uint32_t nv = (uint32_t)min((unsigned long long)vnnz - rbase, (unsigned long long)kBinMaxSlices * E);
        const uint32_t S = (nv + E - 1u) / E;
        const uint32_t tag = gen0 + rounds;
        unsigned long long *slices = a.slices + (size_t)(rounds & 1u) * kBinMaxSlices;
        const gl_idx_val *vec = a.vec + 1u + rbase;
        // ---- 1. the non-zeros in the columns of my slices
        for (uint32_t sl = blk; sl < S; sl += G) {
            unsigned long long deg = 0ull;
            const uint32_t e = sl * E + tid;
            if (tid < E && e < nv) {
                const uint32_t col = vec[e].index;
                if (col < a.num_cols) deg = a.indptr[col + 1u] - a.indptr[col];
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) deg += __shfl_down(deg, d);
            if ((tid & 63u) == 0u) L.wave64[tid >> 6] = deg;
            __syncthreads();
            if (tid == 0) {
                unsigned long long sum = 0ull;
                for (uint32_t k = 0; k < 16u; k++) sum += L.wave64[k];
                const unsigned long long word = ((unsigned long long)tag << 32) | (sum > 0xfffffffeull ? 0xffffffffull : sum);
                __hip_atomic_store(&slices[sl], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        GL_STAMP(0, 1);  // slice sums published
        // ---- 2. everybody's sums: total, overflow, my range's first slice.  ONE wavefront per workgroup polls the tagged
        // words (sixteen wavefronts polling 250 words each queued up at the handful of lines the words live in: 5 us from the
        // last publication to the release; one counter that every workgroup adds to and polls is slower still, same-box
        // 8 us against 4) and scans them in registers; the results reach the other wavefronts through LDS.
        if ((tid >> 6) == GL_POLL_WAVE) {
            const uint32_t lane = tid & 63u;
            unsigned long long P = 0ull, pre0 = 0ull, lo = 0ull, Q = 0ull;
            uint32_t s0 = 0xffffffffu;
            bool over = false;
            for (uint32_t pass = 0; pass < 2u; pass++) {       // pass 0: the total; pass 1: the slice my range starts in
                unsigned long long run = 0ull;
                for (uint32_t q0 = 0; q0 < S; q0 += 64u) {
                    const uint32_t sl = q0 + lane;
                    unsigned long long v = 0ull;
                    if (sl < S) {
                        unsigned long long w;
                        uint32_t spins = 0;
                        while ((uint32_t)((w = __hip_atomic_load(&slices[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag && ++spins < kSpinLimit)
                            __builtin_amdgcn_s_sleep(1);
                        if ((uint32_t)(w >> 32) != tag) {   // gave up: the word belongs to another round -- no products from it, and the run is marked failed
                            __hip_atomic_store(&a.sync[kSyncErr], gen0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            w = 0ull;
                        }
                        v = w & 0xffffffffull;
                        if (v == 0xffffffffull) {
                            over = true;
                            v = 1ull << 40;
                        }
                    }
                    unsigned long long incl = v;
#pragma unroll
                    for (uint32_t dlt = 1; dlt < 64; dlt <<= 1) {
                        const unsigned long long up = __shfl_up(incl, dlt);
                        if (lane >= dlt) incl += up;
                    }
                    if (pass == 1u) {
                        const unsigned long long pre = run + incl - v;
                        const unsigned long long hit = __ballot(pre <= lo && lo < pre + v);
                        if (hit) {
                            const int src = __ffsll((unsigned long long)hit) - 1;
                            s0 = q0 + (uint32_t)src;
                            pre0 = __shfl(pre, src);
                        }
                    }
                    run += __shfl(incl, 63);
                    if (s0 != 0xffffffffu) break;
                }
                if (pass == 0u) {
                    P = run;
                    over = __any(over) || P > 0xfffffffeull;
                    // equal ranges of products, at least 2048 (a batch that small is mostly fixed cost), whole groups of 64
                    Q = (max((P + G - 1u) / G, 2048ull) + 63ull) & ~63ull;
                    lo = (unsigned long long)blk * Q;
                    if (over || lo >= P) break;
                }
            }
            if (lane == 0) {
                L.r_P = P;
                L.r_lo = lo;
                L.r_hi = min(lo + Q, P);
                L.pre0 = pre0;
                L.word = s0;
                L.r_over = over ? 1u : 0u;
            }
        }
        __syncthreads();
        GL_STAMP(0, 2);  // rendezvous over
        const unsigned long long P = L.r_P, lo = L.r_lo, hi = L.r_hi, pre0 = L.pre0;
        const uint32_t s0 = L.word;
        const bool over = L.r_over != 0u;
        if (over) {
            // ---- a vector that names columns again and again (the total does not fit 32 bits): every workgroup takes its own
            // slices, column by column -- slow, never wrong
            for (uint32_t sl = blk; sl < S; sl += G) {
                uint32_t start = 0, deg = 0;
                float xv = 0.0f;
                const uint32_t e = sl * E + tid;
                if (tid < E && e < nv) {
                    const gl_idx_val iv = vec[e];
                    if (iv.index < a.num_cols) {
                        start = a.indptr[iv.index];
                        deg = a.indptr[iv.index + 1u] - start;
                        xv = iv.val;
                    }
                }
                __syncthreads();
                L.start[tid] = start;
                L.pref[tid] = deg;
                L.val[tid] = xv;
                __syncthreads();
                for (uint32_t j = 0; j < E; j++) {
                    const uint32_t dj = L.pref[j], sj = L.start[j];
                    const float xj = L.val[j];
                    for (uint32_t c = 0; c < dj; c += kBinBatch) {
                        __syncthreads();
                        bin_chunk<OP>(a, L, sj + c, min(kBinBatch, dj - c), xj);
                    }
                }
            }
            __syncthreads();
            continue;
        }
        // (equal ranges of products, at least 2048 -- a batch that small is mostly fixed cost --, whole groups of 64)
        if (lo >= P) {                    // block-uniform: nothing left for this workgroup
            __syncthreads();              // (the next round's rendezvous rewrites the results in LDS)
            continue;
        }
        unsigned long long cur = pre0;                // products in front of the staged window
        for (uint32_t ebase = s0 * E; cur < hi && ebase < nv; ebase += kBinSlice) {
            cur += bin_window<OP>(a, L, vec + ebase, min(kBinSlice, nv - ebase), cur, lo, hi, 0u);
            __syncthreads();
        }
        __syncthreads();
    }
    GL_STAMP(0, 10);
    // (the fold kernel that follows advances the generation: spmspv_bin_rounds)
}

// rounds (rendezvous generations) a bin launch of `grid` workgroups uses on a vector of vnnz entries
__host__ __device__ inline uint32_t spmspv_bin_rounds(uint32_t vnnz, uint32_t grid) {
    const uint32_t E = min(kBinSlice, max(1u, (vnnz + grid - 1u) / grid));
    const unsigned long long per = (unsigned long long)kBinMaxSlices * E;
    return (uint32_t)max(1ull, ((unsigned long long)vnnz + per - 1ull) / per);
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
    uint32_t *state;          // tiles words: (generation & 0xffff) << 16 | entries of the tile, once it has counted
    uint32_t *sync;
    uint32_t tickets;         // more tiles than resident workgroups: tiles are handed out in arrival order
    uint32_t merge_all;       // every tile also takes what the dense accumulator holds
    const uint32_t *mode;     // non-null: the same when mode[0] != 0 (the run went row-wise)
    const gl_idx_val *bin_vec;      // the bin launch in front of this one: its vector and grid (the generation it used up)
    uint32_t bin_grid;
    unsigned long long *host_rec;   // page-locked host word: seq << 32 | count when everything has been written (or null)
    uint32_t seq;
};

template <int OPX>
__global__ __launch_bounds__(kFoldThreads) void spmspv_fold_kernel(FoldArgs a) {
    using TL = Tile<OPX>;
    using T = typename TL::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char fold_lds_raw[];
    T *tile = reinterpret_cast<T *>(fold_lds_raw);
    __shared__ unsigned long long s_ball[kFoldMaxRounds * kFoldWaves];   // keep-ballot of (round, wavefront)
    __shared__ uint32_t s_word;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr bool BITS = OPX >= 3;   // the integer value types compare bit patterns (zero may be a NaN as a float)
    const uint32_t T_ = a.tiles.count, R = a.tiles.rows;
    uint32_t t = blockIdx.x;
    GL_STAMP(1, 0);
    if (a.tickets) {   // (every run takes exactly T_ tickets: the counter modulo T_ is the tile, in arrival order)
        if (tid == 0)
            s_word = (uint32_t)(__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.sync + kSyncFoldTicket), 1ull, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT) % T_);
        __syncthreads();
        t = s_word;
    }
    // the tag of this run's tile states: every run rewrites every state, so the previous run's tag is all a reader can meet
    const uint32_t gen0 = a.sync[kSyncGen], tag = (gen0 & 0xffffu) << 16;
    const uint32_t row0 = t * R;
    const uint32_t rows = a.nrows > row0 ? min(R, a.nrows - row0) : 0u;
    const uint32_t rounds = (rows + kFoldThreads - 1u) / kFoldThreads;
    const uint32_t raw = a.cursor[t];
    const uint32_t bb = a.bin_base[t], cap = a.bin_base[t + 1u] - bb;
    const uint32_t cnt = min(raw, cap);
    const bool merge = a.merge_all != 0u || (a.mode && a.mode[0]) || raw > cap;
    uint32_t total = 0, pre_c[4] = {0u, 0u, 0u, 0u};
    if (cnt || merge) {
        for (uint32_t i = tid; i < rows; i += kFoldThreads) tile[i] = TL::ident();
        __syncthreads();
        if (tid == 0 && raw) a.cursor[t] = 0u;   // (behind the barrier: every wavefront has read it)
        GL_STAMP(1, 1);  // cursor read, tile cleared
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
        GL_STAMP(1, 2);  // bin accumulated
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
        // entries in front of every (round, wavefront) ballot: each wavefront scans the at most 256 counts itself, four per lane
        // (index 64 q + lane) -- one barrier instead of three
        {
            const uint32_t nb = rounds * kFoldWaves;
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) {
                if (64u * q >= nb) break;      // (tiles of <= 4096 rows have 64 ballots: one of the four scans; block-uniform)
                const uint32_t i = 64u * q + lane;
                const uint32_t c = i < nb ? (uint32_t)__popcll(s_ball[i]) : 0u;
                uint32_t incl = c;
#pragma unroll
                for (uint32_t dlt = 1; dlt < 64; dlt <<= 1) {
                    const uint32_t up = __shfl_up(incl, dlt);
                    if (lane >= dlt) incl += up;
                }
                pre_c[q] = total + incl - c;
                total += __shfl(incl, 63);
            }
        }
    }
    // ---- publish the tile's count; entries of the tiles in front = this tile's place in the list.  The first wavefront
    // polls the states in front (tiles handed out in arrival order have all started: nobody waits for a tile that has not).
    uint32_t before = 0;
    GL_STAMP(1, 3);      // count known
    if (tid == 0) __hip_atomic_store(&a.state[t], tag | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == GL_POLL_WAVE) {
        for (uint32_t u = lane; u < t; u += 64u) {
            uint32_t w, spins = 0;
            while (((w = __hip_atomic_load(&a.state[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffff0000u) != tag && ++spins < kSpinLimit)
                __builtin_amdgcn_s_sleep(1);
            if ((w & 0xffff0000u) != tag) {   // gave up on a tile in front: the list cannot be placed -- the run is marked failed
                __hip_atomic_store(&a.sync[kSyncErr], gen0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w = 0u;
            }
            before += w & 0xffffu;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
        if (lane == 0) s_word = before;
    }
    __syncthreads();
    before = s_word;
    GL_STAMP(1, 4);      // counts in front seen
    // ---- the write pass
    const uint32_t row_g0 = a.row_begin + row0;
    if (total) {
        for (uint32_t j = 0; j < rounds; j++) {
            const uint32_t r = j * kFoldThreads + tid;
            const uint32_t bi = j * kFoldWaves + wave;       // this wavefront's ballot of the round
            const unsigned long long b = s_ball[bi];
            const uint32_t sel = (bi >> 6) == 0u ? pre_c[0] : (bi >> 6) == 1u ? pre_c[1] : (bi >> 6) == 2u ? pre_c[2] : pre_c[3];
            const uint32_t off = __shfl(sel, (int)(bi & 63u));
            if ((b >> lane) & 1ull) {
                gl_idx_val item;
                item.index = row_g0 + r;
                item.val = *reinterpret_cast<const float *>(&tile[r]);
                a.out[1u + before + off + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = item;
                if (a.assign) a.assign[item.index] = a.assign_val;   // the entry's own row: nobody else touches it
            }
        }
    }
    if (t == T_ - 1u && tid == 0) {
        // the last tile has seen every other tile's state: the list's length, and fresh tags for the next
        // run (every workgroup of this launch has read the generation by now)
        // (a failed rendezvous -- kSyncErr, set by the bin launch in front or by this tile's own look-back -- leaves an EMPTY list)
        const bool failed = __hip_atomic_load(&a.sync[kSyncErr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen0;
        a.out[0].index = failed ? 0u : before + total;
        a.out[0].val = a.head_val;
        if (failed) __hip_atomic_fetch_add(&a.sync[kSyncFailedRuns], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.sync[kSyncGen], gen0 + (a.bin_vec ? spmspv_bin_rounds(a.bin_vec[0].index, a.bin_grid) : 1u), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    GL_STAMP(1, 5);      // list written
    if (!a.host_rec) return;
    // ---- a blocking caller: the last workgroup to finish stores {sequence, count} to page-locked host memory.  ONE atomic
    // per workgroup: the 64-bit word counts the finished workgroups in its high half and sums their entries in the low one, so
    // the last one to arrive has the list's length in the value the atomic returns (no second trip to read it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's stores have been performed
    __syncthreads();
    if (tid == 0) {
        unsigned long long *done = reinterpret_cast<unsigned long long *>(a.sync + kSyncFoldDone);
        const unsigned long long old = __hip_atomic_fetch_add(done, (1ull << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(old >> 32) == T_ - 1u) {
            uint32_t all = (uint32_t)old + total;
            // every workgroup marked its failure (the run's generation) before its arrival above: count 0xffffffff tells gl_spmspv_wait
            if (__hip_atomic_load(&a.sync[kSyncErr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen0) all = 0xffffffffu;
            __hip_atomic_store(a.host_rec, ((unsigned long long)a.seq << 32) | all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the next run's first arrival is a launch away)
        }
    }
}

}  // namespace gl

#endif  // GL_SPMSPV_BIN_H_
