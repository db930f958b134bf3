// Plan creation on the GPU: the O(nnz) half of gl_spmv_plan_create_ex / bool_plan_build.
//
// The reference formats its matrices on the host, single-threaded (csr2cpsr, io/data_formatter.h:456-534: 6.8 s for
// 16.7 M non-zeros); round 1 of this build did it with OpenMP (2.2 s for the 212 M non-zeros of the orkut
// stand-in).  Here the host keeps only what is O(rows + columns) -- row blocks, hot columns, the packed gather
// order, hub rows, group budgets: the same code for both formatters -- and the device does everything that touches
// every non-zero:
//   * column degrees                      one atomic per entry
//   * column-constant ("pattern") test    two passes, wave per row
//   * the per-block column sort           ONE stable radix sort of all entries by (block, hot?, gather index)
//                                         (rocPRIM; the host sorts each block's records on its own)
//   * the delta-coded cold stream         gaps between neighbouring sorted entries give the dummy entries each one needs
//                                         (255 columns per dummy), ONE inclusive scan (rocPRIM) gives every entry its
//                                         position in its unit's stream, a thread per entry writes it and its dummies in
//                                         the stream's final lane-interleaved order
//   * the run-coded hot stream            (general layout) run starts flagged by comparing neighbouring keys, ONE inclusive scan
//                                         (rocPRIM) numbers the runs, a wave per hot group writes slots, values, run mask and base
//   * the row-packed hot stream           (pattern layout, round 6) hot entries stay OUT of the sort: a wave per row counts the row's
//                                         hot entries, ONE scan gives every row its first record, a wave per row writes its records
// Both formatters produce byte-identical arrays (tests/test_gpu_format.py compares them through gl_spmv_plan_export).
#include <cstring>   // rocPRIM's texture iterator calls memset from host code

#include "gl_spmv_plan.h"

#include <rocprim/rocprim.hpp>

namespace gl {

struct DevCsr {
    uint32_t row_begin = 0, row_end = 0;
    uint64_t nz0 = 0, nnz = 0;
    uint32_t *d_indptr = nullptr;    // rows + 1 offsets, GLOBAL (subtract nz0)
    uint32_t *d_indices = nullptr;   // nnz column ids of the shard
    uint32_t *d_data = nullptr;      // nnz value bits
    uint32_t *d_colbits = nullptr;   // per column value bits (left by fmt_detect_pattern for the emission's diagonal test)
    uint32_t num_cols_colbits = 0;
};

namespace {

struct DevMem {   // scratch that lives for one plan creation
    void *p = nullptr;
    ~DevMem() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        GL_HIP(hipMalloc(&p, bytes ? bytes : 16));
        return GL_OK;
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};

constexpr uint32_t kFmtThreads = 256;

inline unsigned wave_grid(uint64_t rows) {   // wave-per-row kernels: 4 waves per workgroup
    const uint64_t want = (rows + 3) / 4;
    const uint64_t cap = (uint64_t)ctx().num_cus * 32u;
    return (unsigned)std::max<uint64_t>(1, std::min(want, cap));
}

__global__ __launch_bounds__(kFmtThreads) void fmt_degree_kernel(const uint32_t *__restrict__ indices, uint64_t nnz, uint32_t num_cols,
                                                                 uint32_t *__restrict__ deg, uint32_t *__restrict__ bad) {
    for (uint64_t i = (uint64_t)blockIdx.x * kFmtThreads + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * kFmtThreads) {
        const uint32_t c = indices[i];
        if (c < num_cols) atomicAdd(&deg[c], 1u);
        else *bad = 1u;
    }
}

// pass 1 of the column-constant test: any off-diagonal writer leaves its value bits (all writers of a constant
// column agree; pass 2 finds out whether they did)
__global__ __launch_bounds__(kFmtThreads) void fmt_pattern_pass1_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                                        const uint32_t *__restrict__ data, uint32_t row_begin, uint32_t rows,
                                                                        uint64_t nz0, uint32_t num_cols, uint32_t *__restrict__ colbits) {
    const uint32_t lane = threadIdx.x & 63u, nwaves = gridDim.x * (kFmtThreads / 64u);
    for (uint32_t row = blockIdx.x * (kFmtThreads / 64u) + (threadIdx.x >> 6); row < rows; row += nwaves) {
        const uint32_t r = row_begin + row;
        const uint64_t s = indptr[row] - nz0, e = indptr[row + 1] - nz0;
        for (uint64_t i = s + lane; i < e; i += 64u) {
            const uint32_t c = indices[i];
            if (c < num_cols && c != r) {
                const uint32_t bits = data[i];
                if (colbits[c] != bits) colbits[c] = bits;   // hub columns: do not hammer a line that already holds the value
            }
        }
    }
}

// pass 2: every entry equals its column's value, or is the row's one diagonal exception
__global__ __launch_bounds__(kFmtThreads) void fmt_pattern_pass2_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                                        const uint32_t *__restrict__ data, uint32_t row_begin, uint32_t rows,
                                                                        uint64_t nz0, uint32_t num_cols, const uint32_t *__restrict__ colbits,
                                                                        uint32_t *__restrict__ diag_has, uint32_t *__restrict__ diag_val,
                                                                        uint32_t *__restrict__ mismatch, unsigned long long *__restrict__ exceptions) {
    const uint32_t lane = threadIdx.x & 63u, nwaves = gridDim.x * (kFmtThreads / 64u);
    for (uint32_t row = blockIdx.x * (kFmtThreads / 64u) + (threadIdx.x >> 6); row < rows; row += nwaves) {
        const uint32_t r = row_begin + row;
        const uint64_t s = indptr[row] - nz0, e = indptr[row + 1] - nz0;
        uint32_t nexc = 0;
        bool bad = false;
        for (uint64_t i = s + lane; i < e; i += 64u) {
            const uint32_t c = indices[i], bits = data[i];
            if (c >= num_cols) { bad = true; continue; }
            if (colbits[c] == bits) continue;
            if (c != r) { bad = true; continue; }
            nexc++;
            diag_val[row] = bits;
            atomicOr(&diag_has[row >> 5], 1u << (row & 31u));
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) nexc += __shfl_down(nexc, d);
        nexc = __shfl(nexc, 0);
        if (nexc > 1u) bad = true;
        if (__ballot(bad) != 0ull && lane == 0) *mismatch = 1u;
        if (nexc && lane == 0) atomicAdd(exceptions, (unsigned long long)nexc);
    }
}

// ------------------------------------------------------------------------------------------ general / pattern layout
// key = block << (cb + 1) | hot << cb | gather index;  dropped entries get the bit above the block field
template <typename K>
__global__ __launch_bounds__(kFmtThreads) void fmt_keys_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                               const uint32_t *__restrict__ data, uint32_t row_begin, uint32_t rows, uint64_t nz0,
                                                               uint32_t num_cols, const uint32_t *__restrict__ bstart, uint32_t nblocks,
                                                               const uint32_t *__restrict__ colmap, const uint32_t *__restrict__ colbits,
                                                               uint32_t cb, uint32_t bb, K *__restrict__ keys, uint2 *__restrict__ payload,
                                                               bool drop_hot) {
    const uint32_t lane = threadIdx.x & 63u, nwaves = gridDim.x * (kFmtThreads / 64u);
    const K drop = (K)1 << (bb + cb + 1u);
    for (uint32_t row = blockIdx.x * (kFmtThreads / 64u) + (threadIdx.x >> 6); row < rows; row += nwaves) {
        const uint32_t r = row_begin + row;
        const uint64_t s = indptr[row] - nz0, e = indptr[row + 1] - nz0;
        if (s == e) continue;
        // block of this row: last b with bstart[b] <= r
        uint32_t lo = 0, hi = nblocks - 1u;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (bstart[mid] <= r) lo = mid; else hi = mid - 1u;
        }
        const uint32_t b = lo, r0 = bstart[b];
        for (uint64_t i = s + lane; i < e; i += 64u) {
            const uint32_t c = indices[i], v = data[i];
            K key;
            if (c >= num_cols || (colbits && c == r && v != colbits[c])) {
                key = drop;   // the row's diagonal exception lives in diag_val (out-of-range columns were reported earlier)
            } else {
                const uint32_t cm = colmap[c];
                key = ((K)b << (cb + 1u)) | ((K)(cm >> 31) << cb) | (K)(cm & 0x7fffffffu);
                if (drop_hot && (cm >> 31)) key = drop;   // row-packed hot stream: written from the rows themselves (fmt_emit_hot_rows_kernel)
            }
            keys[i] = key;
            payload[i] = make_uint2(r - r0, v);
        }
    }
}

// off[i] = first sorted position whose (key >> cb) >= i, i = 2 * block + hot
template <typename K>
__global__ __launch_bounds__(kFmtThreads) void fmt_bounds_kernel(const K *__restrict__ keys, uint64_t n, uint32_t count, uint32_t cb,
                                                                 unsigned long long *__restrict__ off) {
    const uint32_t i = blockIdx.x * kFmtThreads + threadIdx.x;
    if (i >= count) return;
    const K target = (K)i << cb;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    off[i] = lo;
}

struct UnitDesc {
    uint32_t cold_begin, cold_end, hot_begin, hot_end;   // positions in the sorted arrays
    uint32_t goff, r0, nrows_direct, hub_off;            // first cold group; first row; #rows | direct << 31; first hub_rows slot of the block
    uint32_t nhub, seg, hot_e0, present_off;             // first hot element; first entry of the unit's present list
    uint32_t nhot_groups, pad0, pad1, pad2;              // hot groups incl. the padding up to whole elements
};

// ---- the delta-coded cold stream (gl_spmv_plan.h)
// dummies[i] = dummy entries in front of sorted position i: a cold entry whose predecessor is a cold entry of the same block
// and more than 255 gather indices away needs one per 255 columns of the gap.  (Hot and dropped entries: 0.)
template <typename K>
__global__ __launch_bounds__(kFmtThreads) void fmt_gap_kernel(const K *__restrict__ keys, uint64_t n, uint32_t cb, uint32_t bb, uint32_t *__restrict__ dummies) {
    const K cmask = ((K)1 << cb) - 1, drop = (K)1 << (bb + cb + 1u);
    for (uint64_t i = (uint64_t)blockIdx.x * kFmtThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kFmtThreads) {
        uint32_t d = 0u;
        if (i > 0) {
            const K k = keys[i], kp = keys[i - 1];
            if ((k >> cb) == (kp >> cb) && ((k >> cb) & (K)1) == 0 && !(k & drop)) {      // same block, both cold
                const uint32_t gap = (uint32_t)(k & cmask) - (uint32_t)(kp & cmask);
                if (gap > kColdMaxDelta) d = (gap - 1u) / kColdMaxDelta;
            }
        }
        dummies[i] = d;
    }
}
// a unit that begins inside a block's cold entries (column segments) starts its own stream: no dummies in front of its first entry
__global__ __launch_bounds__(kFmtThreads) void fmt_gap_unit_starts_kernel(const UnitDesc *__restrict__ units, uint32_t nunits, uint32_t *__restrict__ dummies) {
    const uint32_t u = blockIdx.x * kFmtThreads + threadIdx.x;
    if (u < nunits && units[u].cold_begin < units[u].cold_end) dummies[units[u].cold_begin] = 0u;
}

// One workgroup per unit, a thread per cold entry: the entry and the dummies in front of it go to their positions in the unit's
// stream (ahead[i] = inclusive scan of the dummy counts), position q = lane q % 64 of group goff + q / 64; then the padding up
// to whole elements.  Hub rows (their list is per block) spread over 16 private slots picked by the entry's lane.
template <typename K, bool PATTERN>
__global__ __launch_bounds__(kThreads) void fmt_emit_cold_kernel(const K *__restrict__ keys, const uint2 *__restrict__ payload,
                                                                 const uint32_t *__restrict__ ahead, const UnitDesc *__restrict__ units,
                                                                 const uint32_t *__restrict__ hub_rows, uint32_t cb,
                                                                 unsigned char *__restrict__ cold, uint32_t *__restrict__ bases,
                                                                 uint4 *__restrict__ plan_units) {
    __shared__ uint8_t hub_of[kMaxBlockRows + 1];
    constexpr uint32_t G = PATTERN ? kColdGroupsPattern : kColdGroupsGeneral;
    constexpr uint32_t EB = PATTERN ? kColdElemBytesPattern : kColdElemBytesGeneral;
    const UnitDesc u = units[blockIdx.x];
    const uint32_t nrows = u.nrows_direct & 0xffffu;
    const K cmask = ((K)1 << cb) - 1;
    if (u.nhub) {
        for (uint32_t i = threadIdx.x; i < nrows; i += kThreads) hub_of[i] = 0xffu;
        __syncthreads();
        if (threadIdx.x < u.nhub) hub_of[hub_rows[u.hub_off + threadIdx.x]] = (uint8_t)threadIdx.x;
        __syncthreads();
    }
    const uint32_t pad0 = nrows + kHubSlots * u.nhub;   // the block's dummy slots: pad0 + lane
    auto put = [&](uint32_t q, uint32_t idx, uint32_t delta, uint32_t slot, uint32_t val) {
        const uint32_t g = u.goff + q / 64u, lane = q % 64u, k = g % G;
        unsigned char *el = cold + (size_t)(g / G) * EB;
        if (!lane) {
            bases[g] = idx;       // (a group's first entry: its index is the group's base)
            delta = 0u;
        }
        if (PATTERN) {
            reinterpret_cast<uint16_t *>(el)[lane * 8u + k] = (uint16_t)slot;
            el[1024u + lane * 8u + k] = (unsigned char)delta;
        } else {
            reinterpret_cast<uint16_t *>(el)[lane * 4u + k] = (uint16_t)slot;
            el[512u + lane * 4u + k] = (unsigned char)delta;
            reinterpret_cast<uint32_t *>(el + 768u)[lane * 4u + k] = val;
        }
    };
    const uint32_t m = u.cold_end - u.cold_begin;
    const uint32_t a0 = m ? ahead[u.cold_begin] : 0u;
    for (uint32_t j = threadIdx.x; j < m; j += kThreads) {
        const uint32_t i = u.cold_begin + j;
        const uint32_t idx = (uint32_t)(keys[i] & cmask);
        const uint32_t before = ahead[i] - a0;                             // dummies of the unit up to and including this entry's
        const uint32_t mine = j ? ahead[i] - ahead[i - 1u] : 0u;           // ... of which in front of this entry
        const uint32_t prev = j ? (uint32_t)(keys[i - 1u] & cmask) : idx;
        const uint32_t q = j + before;
        for (uint32_t t = 1; t <= mine; t++) {
            const uint32_t qd = q - mine + (t - 1u);
            put(qd, prev + kColdMaxDelta * t, kColdMaxDelta, pad0 + qd % 64u, 0u);
        }
        const uint2 pl = payload[i];
        uint32_t slot = pl.x;
        if (u.nhub) {
            const uint32_t hb = hub_of[pl.x];
            if (hb != 0xffu) slot = nrows + kHubSlots * hb + ((q % 64u) & (kHubSlots - 1u));
        }
        put(q, idx, idx - (prev + kColdMaxDelta * mine), slot, pl.y);
    }
    const uint32_t total = m ? m + (ahead[u.cold_end - 1u] - a0) : 0u;
    const uint32_t qend = (total + 64u * G - 1u) / (64u * G) * (64u * G);
    const uint32_t last = m ? (uint32_t)(keys[u.cold_end - 1u] & cmask) : 0u;
    for (uint32_t q = total + threadIdx.x; q < qend; q += kThreads)      // padding: delta 0 behind the last entry, base 0 in all-padding groups
        put(q, (q / 64u == (total - 1u) / 64u && total) ? last : 0u, 0u, pad0 + q % 64u, 0u);
    if (threadIdx.x == 0) {
        plan_units[3u * blockIdx.x] = make_uint4(u.goff, qend / 64u, u.r0, u.nrows_direct);
        plan_units[3u * blockIdx.x + 1u] = make_uint4(u.hub_off, u.nhub, u.nhot_groups, u.seg);
    }
}

// ---- the run-coded hot stream (gl_spmv_plan.h)
// flags[i] = 1 where sorted position i is a hot entry that starts a run: its key differs from its predecessor's (block,
// hot bit or slot).  Dropped entries (the bit above the block field) and cold ones carry hot bit 0.
template <typename K>
__global__ __launch_bounds__(kFmtThreads) void fmt_run_flags_kernel(const K *__restrict__ keys, uint64_t n, uint32_t cb, uint32_t *__restrict__ flags) {
    for (uint64_t i = (uint64_t)blockIdx.x * kFmtThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kFmtThreads) {
        const K k = keys[i];
        flags[i] = (((k >> cb) & (K)1) != 0 && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
    }
}
// a unit that begins inside a run (column segments cut the hot entries by position) starts one of its own
__global__ __launch_bounds__(kFmtThreads) void fmt_run_unit_starts_kernel(const UnitDesc *__restrict__ units, uint32_t nunits, uint32_t *__restrict__ flags) {
    const uint32_t u = blockIdx.x * kFmtThreads + threadIdx.x;
    if (u < nunits && units[u].hot_begin < units[u].hot_end) flags[units[u].hot_begin] = 1u;
}

// One workgroup per unit, one wave per hot group.  runs = inclusive scan of the flags: entry i's table slot within the unit
// is runs[i] - runs[hot_begin].
template <typename K>
__global__ __launch_bounds__(kThreads) void fmt_emit_hot_kernel(const K *__restrict__ keys, const uint2 *__restrict__ payload,
                                                                const uint32_t *__restrict__ runs, const UnitDesc *__restrict__ units,
                                                                const uint32_t *__restrict__ hub_rows, uint32_t cb,
                                                                unsigned char *__restrict__ hot, uint32_t *__restrict__ hot_hdr,
                                                                uint16_t *__restrict__ present, uint4 *__restrict__ plan_units) {
    __shared__ uint8_t hub_of[kMaxBlockRows + 1];
    constexpr uint32_t HG = kHotGroupsGeneral;       // (general layout only: pattern plans carry the row-packed stream, below)
    constexpr uint32_t EB = kHotElemBytesGeneral;
    const UnitDesc u = units[blockIdx.x];
    const uint32_t nrows = u.nrows_direct & 0xffffu;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const K cmask = ((K)1 << cb) - 1;
    if (u.nhub) {
        for (uint32_t i = threadIdx.x; i < nrows; i += kThreads) hub_of[i] = 0xffu;
        __syncthreads();
        if (threadIdx.x < u.nhub) hub_of[hub_rows[u.hub_off + threadIdx.x]] = (uint8_t)threadIdx.x;
        __syncthreads();
    }
    const uint32_t m = u.hot_end - u.hot_begin, nraw = (m + 63u) / 64u;
    const uint32_t s0 = m ? runs[u.hot_begin] : 0u;
    const uint32_t pad_slot = nrows + kHubSlots * u.nhub + lane;   // padding: the lane's dummy slot of the block
    for (uint32_t hg = wave; hg < u.nhot_groups; hg += kWaves) {
        const uint32_t st = u.hot_begin + hg * 64u, cnt = hg < nraw ? min(64u, u.hot_end - st) : 0u;
        const size_t e = (size_t)u.hot_e0 + hg / HG;
        const uint32_t k = hg % HG;
        uint32_t local = 0u;
        bool start = false;
        if (lane >= cnt) {
            reinterpret_cast<uint16_t *>(hot + e * EB)[lane * 4u + k] = (uint16_t)pad_slot;   // (values stay 0)
        } else {
            const uint32_t i = st + lane, si = runs[i];
            local = si - s0;
            start = i == u.hot_begin || runs[i - 1u] != si;
            const uint32_t slot_col = (uint32_t)(keys[i] & cmask);
            if (start) present[u.present_off + local] = (uint16_t)slot_col;
            const uint2 pl = payload[i];
            uint32_t slot = pl.x;
            if (u.nhub) {
                const uint32_t hb = hub_of[pl.x];
                if (hb != 0xffu) slot = nrows + kHubSlots * hb + (lane & (kHubSlots - 1u));
            }
            unsigned char *el = hot + e * EB;
            reinterpret_cast<uint16_t *>(el)[lane * 4u + k] = (uint16_t)slot;
            reinterpret_cast<uint32_t *>(el + 512)[lane * 4u + k] = pl.y;
        }
        const unsigned long long mask = __ballot(start && lane >= 1u) >> 1;   // bit l - 1: entry l starts a run
        const uint32_t base = __shfl(local, 0);
        if (lane == 0) {
            uint32_t *hd = hot_hdr + e * (kHotHdrWordsPerGroup * HG);
            hd[2u * k] = (uint32_t)mask;
            hd[2u * k + 1u] = (uint32_t)(mask >> 32);
            hd[2u * HG + k] = base;
        }
    }
    if (threadIdx.x == 0)
        plan_units[3u * blockIdx.x + 2u] = make_uint4(u.hot_e0, u.present_off, m ? runs[u.hot_end - 1u] - s0 + 1u : 0u, 0u);
}

// ---- the ROW-PACKED hot stream of pattern plans (gl_spmv_plan.h)
// counts[row] = hot entries of the row << 32 | its records (groups of <= 7 hot entries); a wave per row
__global__ __launch_bounds__(kFmtThreads) void fmt_hot_row_counts_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                                         const uint32_t *__restrict__ data, uint32_t row_begin, uint32_t rows,
                                                                         uint64_t nz0, uint32_t num_cols, const uint32_t *__restrict__ colmap,
                                                                         const uint32_t *__restrict__ colbits, unsigned long long *__restrict__ counts) {
    const uint32_t lane = threadIdx.x & 63u, nwaves = gridDim.x * (kFmtThreads / 64u);
    for (uint32_t row = blockIdx.x * (kFmtThreads / 64u) + (threadIdx.x >> 6); row < rows; row += nwaves) {
        const uint32_t r = row_begin + row;
        const uint64_t s = indptr[row] - nz0, e = indptr[row + 1] - nz0;
        uint32_t h = 0;
        for (uint64_t i = s + lane; i < e; i += 64u) {
            const uint32_t c = indices[i];
            if (c >= num_cols || (colbits && c == r && data[i] != colbits[c])) continue;
            h += colmap[c] >> 31;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) h += __shfl_xor(h, d);
        if (lane == 0) counts[row] = ((unsigned long long)h << 32) | (unsigned long long)((h + kHotRecEntries - 1u) / kHotRecEntries);
    }
}
// out[b] = the scanned counts in front of block b's first row (incl[] = INCLUSIVE scan of counts[]; b = nblocks: everything)
__global__ __launch_bounds__(kFmtThreads) void fmt_hot_block_starts_kernel(const unsigned long long *__restrict__ incl, const uint32_t *__restrict__ bstart,
                                                                           uint32_t nblocks, uint32_t row_begin, unsigned long long *__restrict__ out) {
    const uint32_t b = blockIdx.x * kFmtThreads + threadIdx.x;
    if (b > nblocks) return;
    const uint32_t row = bstart[b] - row_begin;
    out[b] = row ? incl[row - 1u] : 0ull;
}
// every field of the unit's elements starts as padding -- the table's identity slot, the lane's dummy row slot -- and the unit's
// third descriptor; one workgroup per unit
__global__ __launch_bounds__(kThreads) void fmt_fill_hot_rows_kernel(const UnitDesc *__restrict__ units, unsigned char *__restrict__ hot, uint32_t nhot_table,
                                                                     uint4 *__restrict__ plan_units) {
    const UnitDesc u = units[blockIdx.x];
    const uint32_t nrows = u.nrows_direct & 0xffffu, pad0 = nrows + kHubSlots * u.nhub;
    const uint32_t ident = nhot_table | (nhot_table << 16);
    for (uint32_t i = threadIdx.x; i < u.nhot_groups * 64u; i += kThreads) {     // (nhot_groups: the unit's hot ELEMENTS here)
        uint4 rec = make_uint4(ident, ident, ident, (nhot_table & 0xffffu) | ((pad0 + (i & 63u)) << 16));
        reinterpret_cast<uint4 *>(hot + (size_t)u.hot_e0 * kHotElemBytesRows)[i] = rec;
    }
    if (threadIdx.x == 0) plan_units[3u * blockIdx.x + 2u] = make_uint4(u.hot_e0, 0u, nhot_table, 0u);
}
// a wave per row: the row's hot entries, in CSR order, go to fields 0..6 of its records; record j of a block sits in the unit
// whose cut [M s / S, M (s + 1) / S) holds j, at lane (j - cut) / chunk of element (j - cut) % chunk (chunk = the unit's elements)
__global__ __launch_bounds__(kFmtThreads) void fmt_emit_hot_rows_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                                        const uint32_t *__restrict__ data, uint32_t row_begin, uint32_t rows,
                                                                        uint64_t nz0, uint32_t num_cols, const uint32_t *__restrict__ colmap,
                                                                        const uint32_t *__restrict__ colbits, const uint32_t *__restrict__ bstart,
                                                                        uint32_t nblocks, const unsigned long long *__restrict__ incl,
                                                                        const unsigned long long *__restrict__ block_start,
                                                                        const uint32_t *__restrict__ seg, const uint32_t *__restrict__ unit_of,
                                                                        const UnitDesc *__restrict__ units, const uint32_t *__restrict__ hub_rows,
                                                                        unsigned char *__restrict__ hot) {
    const uint32_t lane = threadIdx.x & 63u, nwaves = gridDim.x * (kFmtThreads / 64u);
    for (uint32_t row = blockIdx.x * (kFmtThreads / 64u) + (threadIdx.x >> 6); row < rows; row += nwaves) {
        const uint32_t r = row_begin + row;
        const uint64_t s = indptr[row] - nz0, e = indptr[row + 1] - nz0;
        const unsigned long long before = row ? incl[row - 1u] : 0ull, mine = incl[row] - before;
        if ((uint32_t)(mine >> 32) == 0u) continue;    // no hot entry in this row
        uint32_t lo = 0, hi = nblocks - 1u;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (bstart[mid] <= r) lo = mid; else hi = mid - 1u;
        }
        const uint32_t b = lo, row_local = r - bstart[b];
        const uint64_t M = (uint32_t)block_start[b + 1u] - (uint32_t)block_start[b];        // records of the block (low halves)
        const uint64_t rec0 = (uint32_t)before - (uint32_t)block_start[b];                   // the row's first record within the block
        const uint32_t S = seg[b];
        const UnitDesc u0 = units[unit_of[b]];
        const uint32_t nrows = u0.nrows_direct & 0xffffu;
        int hub = -1;
        for (uint32_t k = 0; k < u0.nhub; k++)
            if (hub_rows[u0.hub_off + k] == row_local) hub = (int)k;
        uint32_t running = 0;
        for (uint64_t i0 = s; i0 < e; i0 += 64u) {
            const uint64_t i = i0 + lane;
            uint32_t cm = 0;
            bool is_hot = false;
            if (i < e) {
                const uint32_t c = indices[i];
                if (c < num_cols && !(colbits && c == r && data[i] != colbits[c])) {
                    cm = colmap[c];
                    is_hot = (cm >> 31) != 0u;
                }
            }
            const unsigned long long bal = __ballot(is_hot);
            const uint32_t rank = running + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            running += (uint32_t)__popcll(bal);
            if (!is_hot) continue;
            const uint64_t j = rec0 + rank / kHotRecEntries;
            const uint32_t f = rank % kHotRecEntries;
            uint64_t guess = S == 1u ? 0u : (j * S + S - 1u) / M;
            uint32_t sg = (uint32_t)(guess < S ? guess : S - 1u);
            while (M * sg / S > j) sg--;
            while (M * (sg + 1u) / S <= j) sg++;
            const UnitDesc u = units[unit_of[(size_t)sg * nblocks + b]];
            const uint32_t jj = (uint32_t)(j - M * sg / S), chunk = u.nhot_groups, l = jj / chunk, el = jj % chunk;
            uint16_t *rec = reinterpret_cast<uint16_t *>(hot + (size_t)(u.hot_e0 + el) * kHotElemBytesRows) + l * 8u;
            rec[f] = (uint16_t)(cm & 0x7fffffffu);
            if (f == 0u) rec[7] = (uint16_t)(hub < 0 ? row_local : nrows + kHubSlots * (uint32_t)hub + (l & (kHubSlots - 1u)));
        }
    }
}

// the element of slack behind the stream (the kernel's clamped loads land there, never accumulated): kRowPad in every slot;
// values start as 0 everywhere
__global__ void fmt_fill_hot_kernel(uint32_t *hot, size_t elems, uint32_t words_per_elem, uint32_t slot_words) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < elems * words_per_elem; i += (size_t)gridDim.x * 256u)
        hot[i] = (i % words_per_elem) < slot_words ? (kRowPad | (kRowPad << 16)) : 0u;
}

__global__ void fmt_fill_u32_kernel(uint32_t *dst, uint32_t v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) dst[i] = v;
}

template <typename K>
int sort_pairs(K *kin, K *kout, uint2 *vin, uint2 *vout, uint64_t n, uint32_t bits, hipStream_t s) {
    size_t tmp_bytes = 0;
    GL_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0u, bits, s));
    DevMem tmp;
    int rc = tmp.alloc(tmp_bytes);
    if (rc != GL_OK) return rc;
    GL_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0u, bits, s));
    GL_HIP(hipStreamSynchronize(s));   // tmp dies here
    return GL_OK;
}

template <typename K>
int sort_keys_values_u32(K *kin, K *kout, uint32_t *vin, uint32_t *vout, uint64_t n, uint32_t bits, hipStream_t s) {
    size_t tmp_bytes = 0;
    GL_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0u, bits, s));
    DevMem tmp;
    int rc = tmp.alloc(tmp_bytes);
    if (rc != GL_OK) return rc;
    GL_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0u, bits, s));
    GL_HIP(hipStreamSynchronize(s));
    return GL_OK;
}

inline uint32_t bits_for(uint64_t count) {   // bits needed to hold values 0 .. count - 1
    uint32_t b = 1;
    while ((1ull << b) < count) b++;
    return b;
}

template <typename K>
int emit_general_typed(DevCsr *c, const EmitGeneral &e, gl_spmv_plan p, std::vector<uint32_t> &hub_count, uint64_t *hot_nnz,
                       uint32_t cb, uint32_t bb) {
    hipStream_t s = ctx().stream;
    const BlockPlan &bp = *e.bp;
    const uint32_t nblocks = bp.nblocks, nunits = bp.nunits;
    const uint32_t rows = c->row_end - c->row_begin;
    const uint64_t nnz = c->nnz;
    const bool hot_rows = e.pattern;     // pattern plans: the row-packed hot stream (hot entries stay out of the sort)
    const uint32_t hot_groups = hot_rows ? 1u : kHotGroupsGeneral;
    const uint32_t hot_elem_bytes = hot_rows ? kHotElemBytesRows : kHotElemBytesGeneral;

    DevMem d_bstart, d_colmap, d_keys, d_keys2, d_pl, d_pl2, d_off;
    int rc;
    if ((rc = d_bstart.alloc((size_t)(nblocks + 1) * 4u)) != GL_OK || (rc = d_colmap.alloc((size_t)e.num_cols * 4u)) != GL_OK ||
        (rc = d_keys.alloc(nnz * sizeof(K))) != GL_OK || (rc = d_keys2.alloc(nnz * sizeof(K))) != GL_OK ||
        (rc = d_pl.alloc(nnz * 8u)) != GL_OK || (rc = d_pl2.alloc(nnz * 8u)) != GL_OK ||
        (rc = d_off.alloc((size_t)(2u * nblocks + 1u) * 8u)) != GL_OK)
        return rc;
    GL_HIP(hipMemcpyAsync(d_bstart.p, bp.bstart.data(), (size_t)(nblocks + 1) * 4u, hipMemcpyHostToDevice, s));
    GL_HIP(hipMemcpyAsync(d_colmap.p, e.colmap, (size_t)e.num_cols * 4u, hipMemcpyHostToDevice, s));
    fmt_keys_kernel<K><<<wave_grid(rows), kFmtThreads, 0, s>>>(c->d_indptr, c->d_indices, c->d_data, c->row_begin, rows, c->nz0, e.num_cols,
                                                               d_bstart.as<uint32_t>(), nblocks, d_colmap.as<uint32_t>(),
                                                               e.diag_mode ? c->d_colbits : nullptr, cb, bb, d_keys.as<K>(), d_pl.as<uint2>(), hot_rows);
    GL_LAUNCH_CHECK();
    if ((rc = sort_pairs<K>(d_keys.as<K>(), d_keys2.as<K>(), d_pl.as<uint2>(), d_pl2.as<uint2>(), nnz, bb + cb + 2u, s)) != GL_OK) return rc;
    (void)hipFree(d_keys.p); d_keys.p = nullptr;
    (void)hipFree(d_pl.p); d_pl.p = nullptr;
    const K *keys = d_keys2.as<K>();
    const uint2 *payload = d_pl2.as<uint2>();
    fmt_bounds_kernel<K><<<cdiv(2u * nblocks + 1u, kFmtThreads), kFmtThreads, 0, s>>>(keys, nnz, 2u * nblocks + 1u, cb, d_off.as<unsigned long long>());
    GL_LAUNCH_CHECK();
    std::vector<unsigned long long> off(2u * nblocks + 1u);
    GL_HIP(hipMemcpyAsync(off.data(), d_off.p, off.size() * 8u, hipMemcpyDeviceToHost, s));
    GL_HIP(hipStreamSynchronize(s));

    // ---- host, O(rows): hub rows of every block (same rule as the host formatter) and the unit table
    std::vector<uint32_t> hub_rows((size_t)nblocks * kMaxHubRows, 0u);
    hub_count.assign(nblocks, 0u);
    std::vector<UnitDesc> units(nunits);
    uint64_t hn = 0;
    std::vector<uint64_t> mcs(nblocks), mhs(nblocks), mrec(nblocks, 0);
    for (uint32_t b = 0; b < nblocks; b++) mcs[b] = off[2 * b + 1] - off[2 * b], mhs[b] = off[2 * b + 2] - off[2 * b + 1];
    // ---- row-packed hot stream: hot entries and records per row (a wave per row), ONE scan, the blocks' shares read back
    DevMem d_rowcnt, d_rowincl, d_blockstart;
    if (hot_rows) {
        DevMem d_tmp;
        if ((rc = d_rowcnt.alloc((size_t)std::max(rows, 1u) * 8u)) != GL_OK || (rc = d_rowincl.alloc((size_t)std::max(rows, 1u) * 8u)) != GL_OK ||
            (rc = d_blockstart.alloc((size_t)(nblocks + 1u) * 8u)) != GL_OK)
            return rc;
        fmt_hot_row_counts_kernel<<<wave_grid(rows), kFmtThreads, 0, s>>>(c->d_indptr, c->d_indices, c->d_data, c->row_begin, rows, c->nz0, e.num_cols,
                                                                           d_colmap.as<uint32_t>(), e.diag_mode ? c->d_colbits : nullptr,
                                                                           d_rowcnt.as<unsigned long long>());
        GL_LAUNCH_CHECK();
        size_t tmp_bytes = 0;
        GL_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, d_rowcnt.as<unsigned long long>(), d_rowincl.as<unsigned long long>(), (size_t)rows,
                                       rocprim::plus<unsigned long long>(), s));
        if ((rc = d_tmp.alloc(tmp_bytes)) != GL_OK) return rc;
        GL_HIP(rocprim::inclusive_scan(d_tmp.p, tmp_bytes, d_rowcnt.as<unsigned long long>(), d_rowincl.as<unsigned long long>(), (size_t)rows,
                                       rocprim::plus<unsigned long long>(), s));
        fmt_hot_block_starts_kernel<<<cdiv(nblocks + 1u, kFmtThreads), kFmtThreads, 0, s>>>(d_rowincl.as<unsigned long long>(), d_bstart.as<uint32_t>(), nblocks,
                                                                                            c->row_begin, d_blockstart.as<unsigned long long>());
        GL_LAUNCH_CHECK();
        std::vector<unsigned long long> bs(nblocks + 1u);
        GL_HIP(hipMemcpyAsync(bs.data(), d_blockstart.p, bs.size() * 8u, hipMemcpyDeviceToHost, s));
        GL_HIP(hipStreamSynchronize(s));   // (d_tmp dies here)
        for (uint32_t b = 0; b < nblocks; b++) {
            mhs[b] = (bs[b + 1] >> 32) - (bs[b] >> 32);
            mrec[b] = (uint64_t)((uint32_t)bs[b + 1] - (uint32_t)bs[b]);
        }
    }
    const UnitLayout ul = layout_units(bp, mcs, mhs, e.dummy_max, hot_groups, e.nhot_table, hot_rows ? &mrec : nullptr);
    const uint64_t total_groups = ul.cold_goff[nunits], hot_elems = ul.hot_e0[nunits];
    if (total_groups >= 0xffffffffull || hot_elems >= 0xffffffffull || ul.present_off[nunits] >= 0xffffffffull)
        return set_error(GL_ERR_UNSUPPORTED, "plan creation: more than 2^32 - 1 groups in a shard");
    for (uint32_t b = 0; b < nblocks; b++) {
        const uint32_t r0 = bp.bstart[b], r1 = bp.bstart[b + 1];
        const uint64_t mc = mcs[b], mh = mhs[b], m = mc + mh;
        hn += mh;
        const uint64_t thr = std::max<uint64_t>(256, m / (uint64_t)e.hub_div);
        uint32_t nh = 0;
        for (uint32_t r = r0; r < r1 && nh < kMaxHubRows; r++) {
            uint64_t cnt = (uint64_t)e.h_indptr[r + 1] - e.h_indptr[r];
            if (cnt < thr) continue;   // (a diagonal exception can only lower it)
            if (e.diag_mode) {
                const uint32_t lr = r - c->row_begin;
                if ((e.diag_has[lr >> 5] >> (lr & 31u)) & 1u) cnt--;
            }
            if (cnt >= thr) hub_rows[(size_t)b * kMaxHubRows + nh++] = r - r0;
        }
        hub_count[b] = nh;
        const uint32_t S = bp.seg[b];
        for (uint32_t sg = 0; sg < S; sg++) {
            UnitDesc &u = units[bp.unit_of[sg][b]];
            u.cold_begin = (uint32_t)(off[2 * b] + mc * sg / S);
            u.cold_end = (uint32_t)(off[2 * b] + mc * (sg + 1) / S);
            u.hot_begin = hot_rows ? 0u : (uint32_t)(off[2 * b + 1] + mh * sg / S);     // (row-packed: no hot entries in the sorted arrays)
            u.hot_end = hot_rows ? 0u : (uint32_t)(off[2 * b + 1] + mh * (sg + 1) / S);
            const size_t ui = bp.unit_of[sg][b];
            u.goff = (uint32_t)ul.cold_goff[ui];
            u.r0 = r0;
            u.nrows_direct = (r1 - r0) | (bp.all_direct ? 0x80000000u : 0u);
            u.hub_off = (uint32_t)((size_t)b * kMaxHubRows);
            u.nhub = nh;
            u.seg = sg;
            u.hot_e0 = (uint32_t)ul.hot_e0[ui];
            u.present_off = (uint32_t)ul.present_off[ui];
            u.nhot_groups = (uint32_t)(ul.hot_e0[ui + 1] - ul.hot_e0[ui]) * hot_groups;
            u.pad0 = u.pad1 = u.pad2 = 0u;
        }
    }
    *hot_nnz = hn;

    // ---- outputs, sized and padded exactly like the host formatter's vectors
    const uint32_t cold_groups = e.pattern ? kColdGroupsPattern : kColdGroupsGeneral;
    const size_t cold_elem_bytes = e.pattern ? kColdElemBytesPattern : kColdElemBytesGeneral;
    const size_t entry_bytes = (size_t)(total_groups / cold_groups) * cold_elem_bytes;
    const size_t tail_bytes = cold_elem_bytes;           // one element of slack: the kernels' clamped loads land here
    const size_t n_bases = (size_t)total_groups + 8u;
    GL_HIP(hipMalloc((void **)&p->d_entries, entry_bytes + tail_bytes));
    GL_HIP(hipMalloc((void **)&p->d_bases, n_bases * 4u));
    GL_HIP(hipMalloc((void **)&p->d_units, (size_t)nunits * 3u * sizeof(uint4) + 16u));
    GL_HIP(hipMalloc((void **)&p->d_hub_rows, hub_rows.size() * 4u + 16u));
    // the run-coded hot stream: one element of slack behind it (the kernel's clamped loads), like the host formatter's vectors
    const size_t hot_bytes = (size_t)(hot_elems + 1u) * hot_elem_bytes;
    const size_t hdr_bytes = hot_rows ? (size_t)32 : (size_t)(hot_elems + 1u) * kHotHdrWordsPerGroup * hot_groups * 4u;   // (rows: no headers)
    const size_t present_bytes = (size_t)std::max<uint64_t>(ul.present_off[nunits], 2u) * 2u;
    GL_HIP(hipMalloc((void **)&p->d_hot, hot_bytes));
    GL_HIP(hipMalloc((void **)&p->d_hot_hdr, hdr_bytes));
    GL_HIP(hipMalloc((void **)&p->d_present, present_bytes));
    p->device_bytes += entry_bytes + tail_bytes + n_bases * 4u + (size_t)nunits * 3u * sizeof(uint4) + hub_rows.size() * 4u +
                       hot_bytes + hdr_bytes + present_bytes;
    p->b_entries = entry_bytes + tail_bytes;
    p->b_bases = n_bases * 4u;
    p->b_units = (size_t)nunits * 3u * sizeof(uint4);
    p->b_hub_rows = hub_rows.size() * 4u;
    p->b_hot = hot_bytes;
    p->b_hot_hdr = hdr_bytes;
    p->b_present = present_bytes;
    p->ngroups = total_groups;
    p->nhot_elems = hot_elems;
    GL_HIP(hipMemsetAsync(p->d_hot_hdr, 0, hdr_bytes, s));
    GL_HIP(hipMemsetAsync(p->d_present, 0, present_bytes, s));
    GL_HIP(hipMemsetAsync(p->d_hot, 0, hot_bytes, s));
    fmt_fill_hot_kernel<<<1, 256, 0, s>>>(reinterpret_cast<uint32_t *>(p->d_hot + (size_t)hot_elems * hot_elem_bytes), 1u, hot_elem_bytes / 4u,
                                          hot_rows ? hot_elem_bytes / 4u : hot_groups * 64u * 2u / 4u);
    GL_LAUNCH_CHECK();
    GL_HIP(hipMemsetAsync(p->d_entries, 0, entry_bytes + tail_bytes, s));
    GL_HIP(hipMemsetAsync(p->d_bases, 0, n_bases * 4u, s));
    GL_HIP(hipMemcpyAsync(p->d_hub_rows, hub_rows.data(), hub_rows.size() * 4u, hipMemcpyHostToDevice, s));

    DevMem d_units;
    if ((rc = d_units.alloc((size_t)nunits * sizeof(UnitDesc))) != GL_OK) return rc;
    GL_HIP(hipMemcpyAsync(d_units.p, units.data(), (size_t)nunits * sizeof(UnitDesc), hipMemcpyHostToDevice, s));
    // ---- the cold stream: dummies per gap, one scan, a thread per entry
    {
        DevMem d_gap, d_ahead, d_tmp;
        if ((rc = d_gap.alloc(std::max<uint64_t>(nnz, 1u) * 4u)) != GL_OK || (rc = d_ahead.alloc(std::max<uint64_t>(nnz, 1u) * 4u)) != GL_OK) return rc;
        fmt_gap_kernel<K><<<std::min<unsigned>(cdiv(std::max<uint64_t>(nnz, 1u), kFmtThreads), (unsigned)ctx().num_cus * 32u), kFmtThreads, 0, s>>>(
            keys, nnz, cb, bb, d_gap.as<uint32_t>());
        GL_LAUNCH_CHECK();
        fmt_gap_unit_starts_kernel<<<cdiv(nunits, kFmtThreads), kFmtThreads, 0, s>>>(d_units.as<UnitDesc>(), nunits, d_gap.as<uint32_t>());
        GL_LAUNCH_CHECK();
        size_t tmp_bytes = 0;
        GL_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, d_gap.as<uint32_t>(), d_ahead.as<uint32_t>(), (size_t)nnz, rocprim::plus<uint32_t>(), s));
        if ((rc = d_tmp.alloc(tmp_bytes)) != GL_OK) return rc;
        GL_HIP(rocprim::inclusive_scan(d_tmp.p, tmp_bytes, d_gap.as<uint32_t>(), d_ahead.as<uint32_t>(), (size_t)nnz, rocprim::plus<uint32_t>(), s));
        if (e.pattern)
            fmt_emit_cold_kernel<K, true><<<nunits, kThreads, 0, s>>>(keys, payload, d_ahead.as<uint32_t>(), d_units.as<UnitDesc>(), p->d_hub_rows, cb,
                                                                      reinterpret_cast<unsigned char *>(p->d_entries), p->d_bases, p->d_units);
        else
            fmt_emit_cold_kernel<K, false><<<nunits, kThreads, 0, s>>>(keys, payload, d_ahead.as<uint32_t>(), d_units.as<UnitDesc>(), p->d_hub_rows, cb,
                                                                       reinterpret_cast<unsigned char *>(p->d_entries), p->d_bases, p->d_units);
        GL_LAUNCH_CHECK();
        GL_HIP(hipStreamSynchronize(s));   // scratch dies here
    }
    if (hot_rows) {
        // ---- the row-packed hot stream: padding everywhere, then a wave per row writes its records
        DevMem d_seg, d_unit_of;
        std::vector<uint32_t> unit_of_flat((size_t)bp.Smax * nblocks, 0xffffffffu);
        for (uint32_t sgm = 0; sgm < bp.Smax; sgm++)
            for (uint32_t b = 0; b < nblocks; b++) unit_of_flat[(size_t)sgm * nblocks + b] = bp.unit_of[sgm][b];
        if ((rc = d_seg.alloc((size_t)nblocks * 4u)) != GL_OK || (rc = d_unit_of.alloc(unit_of_flat.size() * 4u)) != GL_OK) return rc;
        GL_HIP(hipMemcpyAsync(d_seg.p, bp.seg.data(), (size_t)nblocks * 4u, hipMemcpyHostToDevice, s));
        GL_HIP(hipMemcpyAsync(d_unit_of.p, unit_of_flat.data(), unit_of_flat.size() * 4u, hipMemcpyHostToDevice, s));
        fmt_fill_hot_rows_kernel<<<nunits, kThreads, 0, s>>>(d_units.as<UnitDesc>(), p->d_hot, e.nhot_table, p->d_units);
        GL_LAUNCH_CHECK();
        if (e.nhot_table) {
            fmt_emit_hot_rows_kernel<<<wave_grid(rows), kFmtThreads, 0, s>>>(c->d_indptr, c->d_indices, c->d_data, c->row_begin, rows, c->nz0, e.num_cols,
                                                                              d_colmap.as<uint32_t>(), e.diag_mode ? c->d_colbits : nullptr,
                                                                              d_bstart.as<uint32_t>(), nblocks, d_rowincl.as<unsigned long long>(),
                                                                              d_blockstart.as<unsigned long long>(), d_seg.as<uint32_t>(),
                                                                              d_unit_of.as<uint32_t>(), d_units.as<UnitDesc>(), p->d_hub_rows, p->d_hot);
            GL_LAUNCH_CHECK();
        }
        GL_HIP(hipStreamSynchronize(s));   // scratch dies here
        p->nhot_lds = e.nhot_table ? e.nhot_table + 64u : 0u;   // the whole table + the identity slots
        return GL_OK;
    }
    // ---- the hot stream: number the runs (one scan over the sorted entries), then a wave per hot group
    {
        DevMem d_flags, d_runs, d_tmp;
        if ((rc = d_flags.alloc(std::max<uint64_t>(nnz, 1u) * 4u)) != GL_OK || (rc = d_runs.alloc(std::max<uint64_t>(nnz, 1u) * 4u)) != GL_OK) return rc;
        fmt_run_flags_kernel<K><<<std::min<unsigned>(cdiv(std::max<uint64_t>(nnz, 1u), kFmtThreads), (unsigned)ctx().num_cus * 32u), kFmtThreads, 0, s>>>(
            keys, nnz, cb, d_flags.as<uint32_t>());
        GL_LAUNCH_CHECK();
        fmt_run_unit_starts_kernel<<<cdiv(nunits, kFmtThreads), kFmtThreads, 0, s>>>(d_units.as<UnitDesc>(), nunits, d_flags.as<uint32_t>());
        GL_LAUNCH_CHECK();
        size_t tmp_bytes = 0;
        GL_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, d_flags.as<uint32_t>(), d_runs.as<uint32_t>(), (size_t)nnz, rocprim::plus<uint32_t>(), s));
        if ((rc = d_tmp.alloc(tmp_bytes)) != GL_OK) return rc;
        GL_HIP(rocprim::inclusive_scan(d_tmp.p, tmp_bytes, d_flags.as<uint32_t>(), d_runs.as<uint32_t>(), (size_t)nnz, rocprim::plus<uint32_t>(), s));
        fmt_emit_hot_kernel<K><<<nunits, kThreads, 0, s>>>(keys, payload, d_runs.as<uint32_t>(), d_units.as<UnitDesc>(), p->d_hub_rows, cb,
                                                           p->d_hot, p->d_hot_hdr, p->d_present, p->d_units);
        GL_LAUNCH_CHECK();
        GL_HIP(hipStreamSynchronize(s));   // scratch dies here
    }
    // the LDS table holds the longest present list
    std::vector<uint4> pu((size_t)nunits * 3u);
    GL_HIP(hipMemcpy(pu.data(), p->d_units, pu.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    uint32_t max_present = 0;
    for (uint32_t u = 0; u < nunits; u++) max_present = std::max(max_present, pu[3u * u + 2u].z);
    p->nhot_lds = (max_present + 63u) / 64u * 64u;
    return GL_OK;
}


// ------------------------------------------------------------------------------------------ (||,&&) layout
// The device twin of bool_plan_build's record loop (gl_spmv_bool.hip): entries whose value is 0 are dropped, the rest
// sorted by (block, column); a group is <= kBoolGroup consecutive entries of ONE x phase whose column stays within
// 2^18 of the group's (word-aligned) base; a span is a run of groups of one phase.
template <typename K>
__global__ __launch_bounds__(kFmtThreads) void fmt_bool_keys_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                                    const uint32_t *__restrict__ data, uint32_t row_begin, uint32_t rows, uint64_t nz0,
                                                                    uint32_t num_cols, const uint32_t *__restrict__ bstart, uint32_t nblocks,
                                                                    uint32_t cb, uint32_t bb, K *__restrict__ keys, uint32_t *__restrict__ payload,
                                                                    uint32_t *__restrict__ row_count, uint32_t *__restrict__ bad) {
    const uint32_t lane = threadIdx.x & 63u, nwaves = gridDim.x * (kFmtThreads / 64u);
    const K drop = (K)1 << (bb + cb);
    for (uint32_t row = blockIdx.x * (kFmtThreads / 64u) + (threadIdx.x >> 6); row < rows; row += nwaves) {
        const uint32_t r = row_begin + row;
        const uint64_t s = indptr[row] - nz0, e = indptr[row + 1] - nz0;
        uint32_t kept = 0;
        if (s != e) {
            uint32_t lo = 0, hi = nblocks - 1u;
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1u) >> 1;
                if (bstart[mid] <= r) lo = mid; else hi = mid - 1u;
            }
            const uint32_t b = lo, r0 = bstart[b];
            for (uint64_t i = s + lane; i < e; i += 64u) {
                const uint32_t c = indices[i];
                if (c >= num_cols) *bad = 1u;
                const bool keep = c < num_cols && __uint_as_float(data[i]) != 0.0f;   // a && b is false for a == 0
                keys[i] = keep ? (((K)b << cb) | (K)c) : drop;
                payload[i] = r - r0;
                kept += keep ? 1u : 0u;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) kept += __shfl_down(kept, d);
        }
        if (lane == 0) row_count[row] = kept;   // entries the row keeps: decides the hub rows
    }
}

struct BoolUnitDesc {
    uint32_t begin, end;          // the unit's piece of the sorted arrays
    uint32_t tmp_goff, tmp_soff;  // where pass A leaves its group / span records
    uint32_t goff, soff;          // final offsets (pass B)
    uint32_t r0, nrows_direct, hub_off, nhub;
    uint32_t pad0, pad1;
};

__device__ __forceinline__ uint32_t bool_phase(uint32_t col) { return col / kBoolPhaseCols; }

// pass A: one wave per unit walks its piece; group records (start, count) and, per group, whether it opens a span
template <typename K>
__global__ __launch_bounds__(64) void fmt_bool_group_kernel(const K *__restrict__ keys, const BoolUnitDesc *__restrict__ units, uint32_t cb,
                                                            uint32_t *__restrict__ gstart, uint32_t *__restrict__ gcount,
                                                            uint2 *__restrict__ counts) {
    const BoolUnitDesc u = units[blockIdx.x];
    const uint32_t lane = threadIdx.x;
    const K cmask = ((K)1 << cb) - 1;
    auto col = [&](uint32_t j) -> uint32_t { return (uint32_t)(keys[j] & cmask); };
    uint32_t cur = u.begin, g = u.tmp_goff, nspans = 0;
    const uint32_t end = u.end;
    while (cur < end) {
        const uint32_t p = cur + kBoolGroup * lane;
        bool full = false;
        uint32_t c0 = 0;
        if (p < end) c0 = col(p);
        if (p < end && end - p >= kBoolGroup) {
            const uint32_t c1 = col(p + kBoolGroup - 1u);
            const uint32_t ph = bool_phase(c0);
            full = bool_phase(c1) == ph && (c1 - ph * kBoolPhaseCols) - ((c0 - ph * kBoolPhaseCols) & ~31u) < (1u << kColOffBits);
        }
        const unsigned long long notfull = __ballot(!full);
        const uint32_t f = notfull ? (uint32_t)__builtin_ctzll(notfull) : 64u;
        // a group opens a span when it is the unit's first or its phase differs from the previous entry's
        const bool mine = lane < f || (lane == f && p < end);
        bool opens = false;
        if (mine) opens = (p == u.begin) || bool_phase(col(p - 1u)) != bool_phase(c0);
        if (lane < f) {
            gstart[g + lane] = p | (opens ? 0x80000000u : 0u);
            gcount[g + lane] = kBoolGroup;
        }
        nspans += (uint32_t)__popcll(__ballot(opens && lane < f));
        g += f;
        if (f == 64u) {
            cur += kBoolGroup * 64u;
            continue;
        }
        const uint32_t pf = cur + kBoolGroup * f;
        if (pf >= end) break;
        const uint32_t cf = col(pf), phf = bool_phase(cf), basef = (cf - phf * kBoolPhaseCols) & ~31u;
        uint32_t cnt = min((uint32_t)kBoolGroup, end - pf);
        for (uint32_t t = 0; t < kBoolGroup / 64u; t++) {
            const uint32_t j = pf + t * 64u + lane;
            bool viol = false;
            if (j < end && j - pf < cnt) {
                const uint32_t cj = col(j);
                viol = bool_phase(cj) != phf || (cj - phf * kBoolPhaseCols) - basef >= (1u << kColOffBits);
            }
            const unsigned long long vm = __ballot(viol);
            if (vm) {
                cnt = t * 64u + (uint32_t)__builtin_ctzll(vm);
                break;
            }
        }
        const bool opens_f = __shfl(opens ? 1 : 0, (int)f) != 0;
        if (lane == 0) {
            gstart[g] = pf | (opens_f ? 0x80000000u : 0u);
            gcount[g] = cnt;
        }
        nspans += opens_f ? 1u : 0u;
        g++;
        cur = pf + cnt;
    }
    if (lane == 0) counts[blockIdx.x] = make_uint2(g - u.tmp_goff, nspans);
}

// pass B: one workgroup per unit; waves emit groups, wave 0 also writes the unit's spans and descriptor
template <typename K>
__global__ __launch_bounds__(kThreads) void fmt_bool_emit_kernel(const K *__restrict__ keys, const uint32_t *__restrict__ payload,
                                                                 const BoolUnitDesc *__restrict__ units, const uint2 *__restrict__ counts,
                                                                 const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ gcount,
                                                                 const uint32_t *__restrict__ hub_rows, uint32_t cb, uint32_t *__restrict__ entries,
                                                                 uint32_t *__restrict__ bases, uint4 *__restrict__ spans, uint4 *__restrict__ plan_units) {
    __shared__ uint8_t hub_of[kMaxBlockRows + 1];
    __shared__ uint32_t s_span_first[kBoolMaxPhases];
    const BoolUnitDesc u = units[blockIdx.x];
    const uint32_t ngroups = counts[blockIdx.x].x, nspans = counts[blockIdx.x].y;
    const uint32_t nrows = u.nrows_direct & 0xffffu;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const K cmask = ((K)1 << cb) - 1;
    if (u.nhub) {
        for (uint32_t i = threadIdx.x; i < nrows; i += kThreads) hub_of[i] = 0xffu;
        __syncthreads();
        if (threadIdx.x < u.nhub) hub_of[hub_rows[u.hub_off + threadIdx.x]] = (uint8_t)threadIdx.x;
        __syncthreads();
    }
    for (uint32_t k = wave; k < ngroups; k += kWaves) {
        const uint32_t st = gstart[u.tmp_goff + k] & 0x7fffffffu, cnt = gcount[u.tmp_goff + k];
        const uint32_t c0 = (uint32_t)(keys[st] & cmask), ph = bool_phase(c0), base = (c0 - ph * kBoolPhaseCols) & ~31u;
        const size_t g = (size_t)u.goff + k;
        for (uint32_t i = lane; i < kBoolGroup; i += 64u) {
            uint32_t e = kRowPad << 5;   // the ghost row slot, column = base
            if (i < cnt) {
                const uint32_t cin = (uint32_t)(keys[st + i] & cmask) - ph * kBoolPhaseCols;
                const uint32_t rl = payload[st + i];
                uint32_t slot = rl;
                if (u.nhub) {
                    const uint32_t hb = hub_of[rl];
                    if (hb != 0xffu) slot = kBoolHubBit0 + (i & (kBoolHubSlots - 1u)) * 32u + hb;
                }
                e = (((cin - base) >> 5) << 19) | (slot << 5) | (cin & 31u);
            }
            entries[g * kBoolGroup + i] = e;
        }
        if (lane == 0) bases[g] = base >> 5;
    }
    if (wave == 0) {
        // spans: the piece is column-sorted, so its phases ascend and it holds at most kBoolMaxPhases spans; collect
        // the groups that open one (64 groups per step), then one lane per span writes it
        volatile uint32_t *first = s_span_first;
        uint32_t sidx = 0;
        for (uint32_t k0 = 0; k0 < ngroups; k0 += 64u) {
            const uint32_t k = k0 + lane;
            const bool opens = k < ngroups && (gstart[u.tmp_goff + k] >> 31);
            const unsigned long long om = __ballot(opens);
            if (opens) {
                const uint32_t my = sidx + (uint32_t)__popcll(om & ((1ull << lane) - 1ull));
                if (my < kBoolMaxPhases) first[my] = k;
            }
            sidx += (uint32_t)__popcll(om);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // one wave: its LDS operations complete in order
        if (lane < nspans && lane < kBoolMaxPhases) {
            const uint32_t k = first[lane], kend = (lane + 1u < nspans) ? first[lane + 1u] : ngroups;
            const uint32_t st = gstart[u.tmp_goff + k] & 0x7fffffffu;
            const uint32_t c0 = (uint32_t)(keys[st] & cmask), ph = bool_phase(c0);
            const uint32_t last_st = gstart[u.tmp_goff + kend - 1u] & 0x7fffffffu, last_cnt = gcount[u.tmp_goff + kend - 1u];
            const uint32_t lo = c0 - ph * kBoolPhaseCols;
            const uint32_t hi = (uint32_t)(keys[last_st + last_cnt - 1u] & cmask) - ph * kBoolPhaseCols;
            spans[(size_t)u.soff + lane] = make_uint4(ph * kBoolPhaseWords, u.goff + k, u.goff + kend, (lo / 128u) | (((hi / 128u) + 1u) << 16));
        }
        if (lane == 0) {
            plan_units[2u * blockIdx.x] = make_uint4(u.soff, nspans, u.r0, u.nrows_direct);
            plan_units[2u * blockIdx.x + 1u] = make_uint4(u.hub_off, u.nhub, 0u, 0u);
        }
    }
}

template <typename K>
int emit_bool_typed(DevCsr *c, const EmitBool &e, gl_spmv_plan p, uint32_t *max_rows_out, uint32_t cb, uint32_t bb) {
    hipStream_t s = ctx().stream;
    const BlockPlan &bp = *e.bp;
    const uint32_t nblocks = bp.nblocks, nunits = bp.nunits;
    const uint32_t rows = c->row_end - c->row_begin;
    const uint64_t nnz = c->nnz;
    const uint32_t nphases = cdiv(e.num_cols, kBoolPhaseCols);

    DevMem d_bstart, d_keys, d_keys2, d_pl, d_pl2, d_off, d_rowcnt;
    int rc;
    if ((rc = d_bstart.alloc((size_t)(nblocks + 1) * 4u)) != GL_OK || (rc = d_keys.alloc(nnz * sizeof(K))) != GL_OK ||
        (rc = d_keys2.alloc(nnz * sizeof(K))) != GL_OK || (rc = d_pl.alloc(nnz * 4u)) != GL_OK || (rc = d_pl2.alloc(nnz * 4u)) != GL_OK ||
        (rc = d_off.alloc((size_t)(nblocks + 2u) * 8u)) != GL_OK || (rc = d_rowcnt.alloc((size_t)std::max(rows, 1u) * 4u)) != GL_OK)
        return rc;
    uint32_t *d_bad = reinterpret_cast<uint32_t *>(d_off.as<unsigned long long>() + nblocks + 1u);
    GL_HIP(hipMemsetAsync(d_bad, 0, 8, s));
    GL_HIP(hipMemcpyAsync(d_bstart.p, bp.bstart.data(), (size_t)(nblocks + 1) * 4u, hipMemcpyHostToDevice, s));
    fmt_bool_keys_kernel<K><<<wave_grid(rows), kFmtThreads, 0, s>>>(c->d_indptr, c->d_indices, c->d_data, c->row_begin, rows, c->nz0, e.num_cols,
                                                                    d_bstart.as<uint32_t>(), nblocks, cb, bb, d_keys.as<K>(), d_pl.as<uint32_t>(),
                                                                    d_rowcnt.as<uint32_t>(), d_bad);
    GL_LAUNCH_CHECK();
    if ((rc = sort_keys_values_u32<K>(d_keys.as<K>(), d_keys2.as<K>(), d_pl.as<uint32_t>(), d_pl2.as<uint32_t>(), nnz, bb + cb + 1u, s)) != GL_OK) return rc;
    (void)hipFree(d_keys.p); d_keys.p = nullptr;
    (void)hipFree(d_pl.p); d_pl.p = nullptr;
    const K *keys = d_keys2.as<K>();
    fmt_bounds_kernel<K><<<cdiv(nblocks + 1u, kFmtThreads), kFmtThreads, 0, s>>>(keys, nnz, nblocks + 1u, cb, d_off.as<unsigned long long>());
    GL_LAUNCH_CHECK();
    std::vector<unsigned long long> off(nblocks + 2u);
    std::vector<uint32_t> rowcnt(rows);
    GL_HIP(hipMemcpyAsync(off.data(), d_off.p, off.size() * 8u, hipMemcpyDeviceToHost, s));
    if (rows) GL_HIP(hipMemcpyAsync(rowcnt.data(), d_rowcnt.p, (size_t)rows * 4u, hipMemcpyDeviceToHost, s));
    GL_HIP(hipStreamSynchronize(s));
    if (off[nblocks + 1u] != 0ull)
        return set_error(GL_ERR_INVALID_ARG, "gl_spmv_plan_create: column index out of range (num_cols %u)", e.num_cols);

    // ---- host, O(rows): hub rows, unit table with scratch budgets
    std::vector<uint32_t> hub_rows((size_t)nblocks * kBoolHubMax, 0u);
    std::vector<BoolUnitDesc> units(nunits);
    uint32_t max_rows = 0;
    uint64_t tmp_g = 0;
    const uint32_t extra = nphases + (e.num_cols >> kColOffBits) + 3u;   // groups a piece can have beyond len / kBoolGroup
    for (uint32_t b = 0; b < nblocks; b++) {
        const uint32_t r0 = bp.bstart[b], r1 = bp.bstart[b + 1];
        max_rows = std::max(max_rows, r1 - r0);
        const uint64_t m = off[b + 1] - off[b];
        const uint64_t thr = std::max<uint64_t>(256, m / 48);
        uint32_t nh = 0;
        for (uint32_t r = r0; r < r1 && nh < kBoolHubMax; r++)
            if (rowcnt[r - c->row_begin] >= thr) hub_rows[(size_t)b * kBoolHubMax + nh++] = r - r0;
        const uint32_t S = bp.seg[b];
        for (uint32_t sg = 0; sg < S; sg++) {
            BoolUnitDesc &u = units[bp.unit_of[sg][b]];
            u.begin = (uint32_t)(off[b] + m * sg / S);
            u.end = (uint32_t)(off[b] + m * (sg + 1) / S);
            u.r0 = r0;
            u.nrows_direct = (r1 - r0) | (bp.all_direct ? 0x80000000u : 0u);
            u.hub_off = (uint32_t)((size_t)b * kBoolHubMax);
            u.nhub = nh;
            u.goff = u.soff = u.tmp_soff = u.pad0 = u.pad1 = 0u;
        }
    }
    for (uint32_t uidx = 0; uidx < nunits; uidx++) {   // scratch offsets in unit order
        units[uidx].tmp_goff = (uint32_t)tmp_g;
        tmp_g += (units[uidx].end - units[uidx].begin + kBoolGroup - 1u) / kBoolGroup + extra;
    }
    if (tmp_g >= 0x7fffffffull) return set_error(GL_ERR_UNSUPPORTED, "gl_spmv_plan_create: too many groups");
    *max_rows_out = max_rows;

    DevMem d_units, d_counts, d_gstart, d_gcount;
    if ((rc = d_units.alloc((size_t)nunits * sizeof(BoolUnitDesc))) != GL_OK || (rc = d_counts.alloc((size_t)nunits * 8u)) != GL_OK ||
        (rc = d_gstart.alloc((size_t)tmp_g * 4u)) != GL_OK || (rc = d_gcount.alloc((size_t)tmp_g * 4u)) != GL_OK)
        return rc;
    GL_HIP(hipMemcpyAsync(d_units.p, units.data(), (size_t)nunits * sizeof(BoolUnitDesc), hipMemcpyHostToDevice, s));
    fmt_bool_group_kernel<K><<<nunits, 64, 0, s>>>(keys, d_units.as<BoolUnitDesc>(), cb, d_gstart.as<uint32_t>(), d_gcount.as<uint32_t>(),
                                                   d_counts.as<uint2>());
    GL_LAUNCH_CHECK();
    std::vector<uint2> counts(nunits);
    GL_HIP(hipMemcpyAsync(counts.data(), d_counts.p, (size_t)nunits * 8u, hipMemcpyDeviceToHost, s));
    GL_HIP(hipStreamSynchronize(s));
    uint64_t total_groups = 0, total_spans = 0;
    for (uint32_t uidx = 0; uidx < nunits; uidx++) {
        units[uidx].goff = (uint32_t)total_groups;
        units[uidx].soff = (uint32_t)total_spans;
        total_groups += counts[uidx].x;
        total_spans += counts[uidx].y;
    }
    if (total_groups >= 0xffffffffull) return set_error(GL_ERR_UNSUPPORTED, "gl_spmv_plan_create: too many groups");
    GL_HIP(hipMemcpyAsync(d_units.p, units.data(), (size_t)nunits * sizeof(BoolUnitDesc), hipMemcpyHostToDevice, s));

    const size_t b_entries = (size_t)total_groups * kBoolGroup * 4u, b_bases = (size_t)total_groups * 4u;
    const size_t b_units = (size_t)nunits * 2u * sizeof(uint4), b_hub = hub_rows.size() * 4u, b_spans = (size_t)total_spans * sizeof(uint4);
    GL_HIP(hipMalloc((void **)&p->d_entries, b_entries ? b_entries : 16));
    GL_HIP(hipMalloc((void **)&p->d_bases, b_bases ? b_bases : 16));
    GL_HIP(hipMalloc((void **)&p->d_units, b_units ? b_units : 16));
    GL_HIP(hipMalloc((void **)&p->d_hub_rows, b_hub ? b_hub : 16));
    GL_HIP(hipMalloc((void **)&p->d_spans, b_spans ? b_spans : 16));
    p->device_bytes += b_entries + b_bases + b_units + b_hub + b_spans;
    p->b_entries = b_entries, p->b_bases = b_bases, p->b_units = b_units, p->b_hub_rows = b_hub, p->b_spans = b_spans;
    if (b_hub) GL_HIP(hipMemcpyAsync(p->d_hub_rows, hub_rows.data(), b_hub, hipMemcpyHostToDevice, s));
    if (nunits) {
        fmt_bool_emit_kernel<K><<<nunits, kThreads, 0, s>>>(keys, d_pl2.as<uint32_t>(), d_units.as<BoolUnitDesc>(), d_counts.as<uint2>(),
                                                            d_gstart.as<uint32_t>(), d_gcount.as<uint32_t>(), p->d_hub_rows, cb,
                                                            reinterpret_cast<uint32_t *>(p->d_entries), p->d_bases, p->d_spans, p->d_units);
        GL_LAUNCH_CHECK();
    }
    GL_HIP(hipStreamSynchronize(s));
    p->ngroups = total_groups;
    return GL_OK;
}

}  // namespace

bool format_on_device(uint32_t flags, uint64_t nnz) {
    if (flags & GL_PLAN_HOST_FORMAT) return false;
    if (flags & GL_PLAN_DEVICE_FORMAT) return true;
    const long forced = env_long("GRAPHLILY_PLAN_DEVICE", -1);
    if (forced >= 0) return forced != 0;
    // below ~1 M entries the host formats faster than the device path's allocations and launches cost
    return nnz >= (1u << 20);
}

int devcsr_stage(DevCsr **out, const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data, uint32_t row_begin,
                 uint32_t row_end) {
    DevCsr *c = new DevCsr();
    c->row_begin = row_begin;
    c->row_end = row_end;
    c->nz0 = h_indptr[row_begin];
    c->nnz = (uint64_t)h_indptr[row_end] - c->nz0;
    const size_t rows1 = (size_t)(row_end - row_begin) + 1u;
    hipError_t e;
    if ((e = hipMalloc((void **)&c->d_indptr, rows1 * 4u)) != hipSuccess ||
        (e = hipMalloc((void **)&c->d_indices, c->nnz ? c->nnz * 4u : 16u)) != hipSuccess ||
        (e = hipMalloc((void **)&c->d_data, c->nnz ? c->nnz * 4u : 16u)) != hipSuccess) {
        devcsr_release(c);
        return set_error(GL_ERR_HIP, "plan creation: staging the CSR on the device: %s", hipGetErrorString(e));
    }
    hipStream_t s = ctx().stream;
    // pageable sources copy at PCIe rate on this platform (profiles/r02_ubench_host.txt: 56 GB/s)
    if ((e = hipMemcpyAsync(c->d_indptr, h_indptr + row_begin, rows1 * 4u, hipMemcpyHostToDevice, s)) != hipSuccess ||
        (c->nnz && (e = hipMemcpyAsync(c->d_indices, h_indices + c->nz0, c->nnz * 4u, hipMemcpyHostToDevice, s)) != hipSuccess) ||
        (c->nnz && (e = hipMemcpyAsync(c->d_data, h_data + c->nz0, c->nnz * 4u, hipMemcpyHostToDevice, s)) != hipSuccess) ||
        (e = hipStreamSynchronize(s)) != hipSuccess) {
        devcsr_release(c);
        return set_error(GL_ERR_HIP, "plan creation: uploading the CSR: %s", hipGetErrorString(e));
    }
    *out = c;
    return GL_OK;
}

// A whole-matrix boolean plan keeps the staged rows for the bottom-up BFS step: entries whose value is zero (a && b is
// false for them) get the column 0xffffffff, which no frontier holds.
__global__ __launch_bounds__(256) void fmt_mark_zero_values_kernel(uint32_t *__restrict__ indices, const uint32_t *__restrict__ data,
                                                                   uint64_t nnz) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * 256u)
        if ((data[i] << 1) == 0u) indices[i] = 0xffffffffu;
}

int devcsr_adopt_rows(DevCsr *c, uint32_t **d_indptr, uint32_t **d_indices) {
    if (c->nnz) {
        fmt_mark_zero_values_kernel<<<(unsigned)std::min<uint64_t>((c->nnz + 255u) / 256u, (uint64_t)ctx().num_cus * 16u), 256, 0, ctx().stream>>>(
            c->d_indices, c->d_data, c->nnz);
        GL_LAUNCH_CHECK();
        GL_HIP(hipStreamSynchronize(ctx().stream));
    }
    *d_indptr = c->d_indptr;
    *d_indices = c->d_indices;
    c->d_indptr = nullptr;
    c->d_indices = nullptr;
    return GL_OK;
}

void devcsr_release(DevCsr *c) {
    if (!c) return;
    (void)hipFree(c->d_indptr);
    (void)hipFree(c->d_indices);
    (void)hipFree(c->d_data);
    (void)hipFree(c->d_colbits);
    delete c;
}

int fmt_column_degrees(DevCsr *c, uint32_t num_cols, std::vector<uint32_t> &deg, int *bad_col) {
    hipStream_t s = ctx().stream;
    DevMem d_deg, d_bad;
    int rc;
    if ((rc = d_deg.alloc((size_t)num_cols * 4u)) != GL_OK || (rc = d_bad.alloc(16)) != GL_OK) return rc;
    GL_HIP(hipMemsetAsync(d_deg.p, 0, (size_t)num_cols * 4u, s));
    GL_HIP(hipMemsetAsync(d_bad.p, 0, 16, s));
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((c->nnz + kFmtThreads - 1) / kFmtThreads, (uint64_t)ctx().num_cus * 16u));
    fmt_degree_kernel<<<grid, kFmtThreads, 0, s>>>(c->d_indices, c->nnz, num_cols, d_deg.as<uint32_t>(), d_bad.as<uint32_t>());
    GL_LAUNCH_CHECK();
    deg.resize(num_cols);
    uint32_t bad = 0;
    GL_HIP(hipMemcpyAsync(deg.data(), d_deg.p, (size_t)num_cols * 4u, hipMemcpyDeviceToHost, s));
    GL_HIP(d2h_word_sync(&bad, d_bad.p, s));
    *bad_col = bad ? 1 : 0;
    return GL_OK;
}

int fmt_detect_pattern(DevCsr *c, uint32_t num_cols, std::vector<uint32_t> &colbits, std::vector<uint32_t> &diag_has,
                       std::vector<float> &diag_val, int *mismatch, uint64_t *exceptions) {
    hipStream_t s = ctx().stream;
    const uint32_t rows = c->row_end - c->row_begin;
    const size_t words = ((size_t)rows + 31u) / 32u;
    if (!c->d_colbits) GL_HIP(hipMalloc((void **)&c->d_colbits, (size_t)std::max(num_cols, 1u) * 4u));
    c->num_cols_colbits = num_cols;
    DevMem d_has, d_val, d_flags;
    int rc;
    if ((rc = d_has.alloc(words * 4u)) != GL_OK || (rc = d_val.alloc((size_t)rows * 4u)) != GL_OK || (rc = d_flags.alloc(16)) != GL_OK) return rc;
    GL_HIP(hipMemsetAsync(c->d_colbits, 0, (size_t)num_cols * 4u, s));
    GL_HIP(hipMemsetAsync(d_has.p, 0, words * 4u, s));
    GL_HIP(hipMemsetAsync(d_val.p, 0, (size_t)rows * 4u, s));
    GL_HIP(hipMemsetAsync(d_flags.p, 0, 16, s));
    fmt_pattern_pass1_kernel<<<wave_grid(rows), kFmtThreads, 0, s>>>(c->d_indptr, c->d_indices, c->d_data, c->row_begin, rows, c->nz0, num_cols, c->d_colbits);
    GL_LAUNCH_CHECK();
    fmt_pattern_pass2_kernel<<<wave_grid(rows), kFmtThreads, 0, s>>>(c->d_indptr, c->d_indices, c->d_data, c->row_begin, rows, c->nz0, num_cols,
                                                                  c->d_colbits, d_has.as<uint32_t>(), d_val.as<uint32_t>(), d_flags.as<uint32_t>(),
                                                                  reinterpret_cast<unsigned long long *>(d_flags.as<uint32_t>() + 2));
    GL_LAUNCH_CHECK();
    colbits.resize(num_cols);
    diag_has.resize(words);
    diag_val.resize(rows);
    uint32_t flags[4] = {0, 0, 0, 0};
    GL_HIP(hipMemcpyAsync(colbits.data(), c->d_colbits, (size_t)num_cols * 4u, hipMemcpyDeviceToHost, s));
    if (words) GL_HIP(hipMemcpyAsync(diag_has.data(), d_has.p, words * 4u, hipMemcpyDeviceToHost, s));
    if (rows) GL_HIP(hipMemcpyAsync(diag_val.data(), d_val.p, (size_t)rows * 4u, hipMemcpyDeviceToHost, s));
    GL_HIP(hipMemcpyAsync(flags, d_flags.p, 16, hipMemcpyDeviceToHost, s));
    GL_HIP(hipStreamSynchronize(s));
    *mismatch = flags[0] ? 1 : 0;
    unsigned long long ex;
    memcpy(&ex, &flags[2], 8);
    *exceptions = ex;
    return GL_OK;
}

int fmt_emit_general(DevCsr *c, const EmitGeneral &e, gl_spmv_plan p, std::vector<uint32_t> &hub_count, uint64_t *hot_nnz) {
    const uint32_t cb = bits_for(std::max<uint64_t>(std::max(e.gather_cols, e.nhot_table), 2u));
    const uint32_t bb = bits_for(std::max<uint32_t>(e.bp->nblocks, 2u));
    if (c->nnz >= 0xffffffffull) return set_error(GL_ERR_UNSUPPORTED, "plan creation on the device: more than 2^32 - 1 entries in a shard");
    if (bb + cb + 2u <= 32u) return emit_general_typed<uint32_t>(c, e, p, hub_count, hot_nnz, cb, bb);
    return emit_general_typed<unsigned long long>(c, e, p, hub_count, hot_nnz, cb, bb);
}

// ------------------------------------------------------------------------------------------ SpMSpV stream, csr2csc, normalise
namespace {

// whole matrix: the CSC arrays simply interleave into the {row, value} stream
__global__ __launch_bounds__(kFmtThreads) void fmt_csc_interleave_kernel(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ vals, uint64_t nnz,
                                                                         uint32_t num_rows, uint2 *__restrict__ stream, uint32_t *__restrict__ bad) {
    for (uint64_t i = (uint64_t)blockIdx.x * kFmtThreads + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * kFmtThreads) {
        const uint32_t r = rows[i];
        if (r >= num_rows) *bad = 1u;
        stream[i] = make_uint2(r, vals[i]);
    }
}

// row shard, pass 1: entries of each column whose row lies in [r0, r1)
__global__ __launch_bounds__(kFmtThreads) void fmt_csc_count_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ rows, uint32_t num_cols,
                                                                    uint32_t num_rows, uint32_t r0, uint32_t r1, uint32_t *__restrict__ cnt,
                                                                    uint32_t *__restrict__ bad) {
    const uint32_t lane = threadIdx.x & 63u, nwaves = gridDim.x * (kFmtThreads / 64u);
    for (uint32_t c = blockIdx.x * (kFmtThreads / 64u) + (threadIdx.x >> 6); c < num_cols; c += nwaves) {
        uint32_t k = 0;
        for (uint64_t i = (uint64_t)indptr[c] + lane; i < indptr[c + 1]; i += 64u) {
            const uint32_t r = rows[i];
            if (r >= num_rows) *bad = 1u;
            k += (r >= r0 && r < r1) ? 1u : 0u;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) k += __shfl_down(k, d);
        if (lane == 0) cnt[c] = k;
    }
}

// pass 2: order-preserving write of the kept entries
__global__ __launch_bounds__(kFmtThreads) void fmt_csc_filter_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ rows,
                                                                     const uint32_t *__restrict__ vals, uint32_t num_cols, uint32_t r0, uint32_t r1,
                                                                     const uint32_t *__restrict__ out_ptr, uint2 *__restrict__ stream) {
    const uint32_t lane = threadIdx.x & 63u, nwaves = gridDim.x * (kFmtThreads / 64u);
    for (uint32_t c = blockIdx.x * (kFmtThreads / 64u) + (threadIdx.x >> 6); c < num_cols; c += nwaves) {
        uint32_t at = out_ptr[c];
        const uint64_t e = indptr[c + 1];
        for (uint64_t i0 = indptr[c]; i0 < e; i0 += 64u) {
            const uint64_t i = i0 + lane;
            uint32_t r = 0;
            bool keep = false;
            if (i < e) {
                r = rows[i];
                keep = r >= r0 && r < r1;
            }
            const unsigned long long km = __ballot(keep);
            if (keep) stream[at + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = make_uint2(r, vals[i]);
            at += (uint32_t)__popcll(km);
        }
    }
}

__global__ __launch_bounds__(kFmtThreads) void fmt_row_of_kernel(const uint32_t *__restrict__ indptr, uint32_t rows, uint32_t *__restrict__ row_of) {
    const uint32_t lane = threadIdx.x & 63u, nwaves = gridDim.x * (kFmtThreads / 64u);
    for (uint32_t r = blockIdx.x * (kFmtThreads / 64u) + (threadIdx.x >> 6); r < rows; r += nwaves)
        for (uint64_t i = (uint64_t)indptr[r] + lane; i < indptr[r + 1]; i += 64u) row_of[i] = r;
}

__global__ __launch_bounds__(kFmtThreads) void fmt_split_pairs_kernel(const uint2 *__restrict__ pairs, uint64_t nnz, uint32_t *__restrict__ a, uint32_t *__restrict__ b) {
    for (uint64_t i = (uint64_t)blockIdx.x * kFmtThreads + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * kFmtThreads) {
        const uint2 p = pairs[i];
        a[i] = p.x;
        b[i] = p.y;
    }
}

__global__ __launch_bounds__(kFmtThreads) void fmt_pack_pairs_kernel(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint64_t nnz, uint2 *__restrict__ pairs) {
    for (uint64_t i = (uint64_t)blockIdx.x * kFmtThreads + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * kFmtThreads) pairs[i] = make_uint2(a[i], b[i]);
}

// adj_data[i] = 1.0 / count(column of i): double divide, float store (io/data_formatter.h:36-51)
__global__ __launch_bounds__(kFmtThreads) void fmt_normalize_kernel(const uint32_t *__restrict__ cols, const uint32_t *__restrict__ deg, uint64_t nnz, float *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * kFmtThreads + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * kFmtThreads)
        out[i] = (float)(1.0 / (double)deg[cols[i]]);
}

inline unsigned flat_grid(uint64_t n) {
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + kFmtThreads - 1) / kFmtThreads, (uint64_t)ctx().num_cus * 16u));
}

}  // namespace

int fmt_spmspv_stream(uint32_t num_rows, uint32_t num_cols, const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data,
                      uint32_t r0, uint32_t r1, uint32_t **d_indptr_out, uint2 **d_stream_out, std::vector<uint32_t> &indptr_out) {
    hipStream_t s = ctx().stream;
    const uint64_t nnz_all = h_indptr[num_cols];
    const bool whole = (r0 == 0 && r1 == num_rows);
    DevMem d_rows, d_vals, d_ip, d_bad, d_cnt;
    int rc;
    if ((rc = d_rows.alloc(nnz_all * 4u)) != GL_OK || (rc = d_vals.alloc(nnz_all * 4u)) != GL_OK || (rc = d_bad.alloc(16)) != GL_OK ||
        (rc = d_ip.alloc((size_t)(num_cols + 1ull) * 4u)) != GL_OK)
        return rc;
    GL_HIP(hipMemsetAsync(d_bad.p, 0, 16, s));
    GL_HIP(hipMemcpyAsync(d_ip.p, h_indptr, (size_t)(num_cols + 1ull) * 4u, hipMemcpyHostToDevice, s));
    if (nnz_all) {
        GL_HIP(hipMemcpyAsync(d_rows.p, h_indices, nnz_all * 4u, hipMemcpyHostToDevice, s));
        GL_HIP(hipMemcpyAsync(d_vals.p, h_data, nnz_all * 4u, hipMemcpyHostToDevice, s));
    }
    uint2 *stream = nullptr;
    uint32_t *out_ip = nullptr;
    struct Guard {   // the two output arrays go back to the driver on every early return
        uint2 *&stream;
        uint32_t *&out_ip;
        bool released = false;
        ~Guard() {
            if (released) return;
            (void)hipFree(stream);
            (void)hipFree(out_ip);
        }
    } guard{stream, out_ip};
    indptr_out.resize((size_t)num_cols + 1u);
    if (whole) {
        GL_HIP(hipMalloc((void **)&stream, nnz_all ? nnz_all * 8u : 16u));
        if (nnz_all) {
            fmt_csc_interleave_kernel<<<flat_grid(nnz_all), kFmtThreads, 0, s>>>(d_rows.as<uint32_t>(), d_vals.as<uint32_t>(), nnz_all, num_rows, stream, d_bad.as<uint32_t>());
            GL_LAUNCH_CHECK();
        }
        out_ip = d_ip.as<uint32_t>();
        d_ip.p = nullptr;   // ownership moves to the caller
        memcpy(indptr_out.data(), h_indptr, indptr_out.size() * 4u);
    } else {
        if ((rc = d_cnt.alloc((size_t)(num_cols + 1ull) * 4u)) != GL_OK) return rc;
        GL_HIP(hipMemsetAsync(d_cnt.p, 0, (size_t)(num_cols + 1ull) * 4u, s));
        fmt_csc_count_kernel<<<wave_grid(num_cols), kFmtThreads, 0, s>>>(d_ip.as<uint32_t>(), d_rows.as<uint32_t>(), num_cols, num_rows, r0, r1,
                                                                     d_cnt.as<uint32_t>(), d_bad.as<uint32_t>());
        GL_LAUNCH_CHECK();
        GL_HIP(hipMalloc((void **)&out_ip, (size_t)(num_cols + 1ull) * 4u));
        size_t tmp_bytes = 0;
        hipError_t he = rocprim::exclusive_scan(nullptr, tmp_bytes, d_cnt.as<uint32_t>(), out_ip, 0u, (size_t)num_cols + 1u, rocprim::plus<uint32_t>(), s);
        DevMem tmp;
        if (he == hipSuccess && (rc = tmp.alloc(tmp_bytes)) == GL_OK)
            he = rocprim::exclusive_scan(tmp.p, tmp_bytes, d_cnt.as<uint32_t>(), out_ip, 0u, (size_t)num_cols + 1u, rocprim::plus<uint32_t>(), s);
        if (he != hipSuccess || rc != GL_OK) {
            return rc != GL_OK ? rc : set_error(GL_ERR_HIP, "gl_spmspv_plan_create: scan: %s", hipGetErrorString(he));
        }
        GL_HIP(hipMemcpyAsync(indptr_out.data(), out_ip, indptr_out.size() * 4u, hipMemcpyDeviceToHost, s));
        GL_HIP(hipStreamSynchronize(s));
        const uint64_t kept = indptr_out[num_cols];
        GL_HIP(hipMalloc((void **)&stream, kept ? kept * 8u : 16u));
        fmt_csc_filter_kernel<<<wave_grid(num_cols), kFmtThreads, 0, s>>>(d_ip.as<uint32_t>(), d_rows.as<uint32_t>(), d_vals.as<uint32_t>(), num_cols, r0, r1,
                                                                      out_ip, stream);
        GL_LAUNCH_CHECK();
    }
    uint32_t bad = 0;
    GL_HIP(d2h_word_sync(&bad, d_bad.p, s));
    if (bad) return set_error(GL_ERR_INVALID_ARG, "gl_spmspv_plan_create: row index out of range (num_rows %u)", num_rows);
    guard.released = true;
    *d_indptr_out = out_ip;
    *d_stream_out = stream;
    return GL_OK;
}

// io::csr2csc (io/data_loader.h:108-144) as ONE stable radix sort by column of (row, value) pairs: rows inside a column
// stay ascending exactly like the reference's row-by-row counting sort
int fmt_csr2csc(uint32_t num_rows, uint32_t num_cols, const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data,
                uint32_t *csc_indptr, uint32_t *csc_indices, float *csc_data) {
    hipStream_t s = ctx().stream;
    const uint64_t nnz = h_indptr[num_rows];
    if (nnz >= 0xffffffffull) return set_error(GL_ERR_UNSUPPORTED, "gl_csr2csc: more than 2^32 - 1 entries");
    // the kernels index the entry arrays with these offsets: a row pointer array that is not monotone (or starts off zero)
    // would send them out of bounds
    if (h_indptr[0] != 0u) return set_error(GL_ERR_INVALID_ARG, "gl_csr2csc: indptr[0] must be 0");
    for (uint32_t r = 0; r < num_rows; r++)
        if (h_indptr[r + 1] < h_indptr[r]) return set_error(GL_ERR_INVALID_ARG, "gl_csr2csc: indptr decreases at row %u", r);
    DevMem d_ip, d_cols, d_cols2, d_rowof, d_vals, d_pairs, d_pairs2, d_deg, d_bad;
    int rc;
    if ((rc = d_ip.alloc((size_t)(num_rows + 1ull) * 4u)) != GL_OK || (rc = d_cols.alloc(nnz * 4u)) != GL_OK || (rc = d_cols2.alloc(nnz * 4u)) != GL_OK ||
        (rc = d_rowof.alloc(nnz * 4u)) != GL_OK || (rc = d_vals.alloc(nnz * 4u)) != GL_OK || (rc = d_pairs.alloc(nnz * 8u)) != GL_OK ||
        (rc = d_pairs2.alloc(nnz * 8u)) != GL_OK || (rc = d_deg.alloc((size_t)(num_cols + 1ull) * 4u)) != GL_OK || (rc = d_bad.alloc(16)) != GL_OK)
        return rc;
    GL_HIP(hipMemcpyAsync(d_ip.p, h_indptr, (size_t)(num_rows + 1ull) * 4u, hipMemcpyHostToDevice, s));
    GL_HIP(hipMemsetAsync(d_deg.p, 0, (size_t)(num_cols + 1ull) * 4u, s));
    GL_HIP(hipMemsetAsync(d_bad.p, 0, 16, s));
    if (nnz) {
        GL_HIP(hipMemcpyAsync(d_cols.p, h_indices, nnz * 4u, hipMemcpyHostToDevice, s));
        GL_HIP(hipMemcpyAsync(d_vals.p, h_data, nnz * 4u, hipMemcpyHostToDevice, s));
        fmt_row_of_kernel<<<wave_grid(num_rows), kFmtThreads, 0, s>>>(d_ip.as<uint32_t>(), num_rows, d_rowof.as<uint32_t>());
        GL_LAUNCH_CHECK();
        fmt_pack_pairs_kernel<<<flat_grid(nnz), kFmtThreads, 0, s>>>(d_rowof.as<uint32_t>(), d_vals.as<uint32_t>(), nnz, d_pairs.as<uint2>());
        GL_LAUNCH_CHECK();
        fmt_degree_kernel<<<flat_grid(nnz), kFmtThreads, 0, s>>>(d_cols.as<uint32_t>(), nnz, num_cols, d_deg.as<uint32_t>(), d_bad.as<uint32_t>());
        GL_LAUNCH_CHECK();
        if ((rc = sort_pairs<uint32_t>(d_cols.as<uint32_t>(), d_cols2.as<uint32_t>(), d_pairs.as<uint2>(), d_pairs2.as<uint2>(), nnz,
                                       bits_for(std::max<uint64_t>(num_cols, 2u)), s)) != GL_OK)
            return rc;
        fmt_split_pairs_kernel<<<flat_grid(nnz), kFmtThreads, 0, s>>>(d_pairs2.as<uint2>(), nnz, d_rowof.as<uint32_t>(), d_vals.as<uint32_t>());
        GL_LAUNCH_CHECK();
    }
    uint32_t bad = 0;   // (read back at once: no early return below may leave a copy into this stack word pending)
    GL_HIP(d2h_word_sync(&bad, d_bad.p, s));
    if (bad) return set_error(GL_ERR_INVALID_ARG, "gl_csr2csc: column index out of range (num_cols %u)", num_cols);
    // column pointers: exclusive scan of the degrees (num_cols + 1 entries, the last one being the total)
    {
        size_t tmp_bytes = 0;
        DevMem d_cp;
        if ((rc = d_cp.alloc((size_t)(num_cols + 1ull) * 4u)) != GL_OK) return rc;
        uint32_t *out = d_cp.as<uint32_t>();
        GL_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, d_deg.as<uint32_t>(), out, 0u, (size_t)num_cols + 1u, rocprim::plus<uint32_t>(), s));
        DevMem tmp;
        if ((rc = tmp.alloc(tmp_bytes)) != GL_OK) return rc;
        GL_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, d_deg.as<uint32_t>(), out, 0u, (size_t)num_cols + 1u, rocprim::plus<uint32_t>(), s));
        GL_HIP(hipMemcpyAsync(csc_indptr, out, (size_t)(num_cols + 1ull) * 4u, hipMemcpyDeviceToHost, s));
        GL_HIP(hipStreamSynchronize(s));
    }
    if (nnz) {
        GL_HIP(hipMemcpyAsync(csc_indices, d_rowof.p, nnz * 4u, hipMemcpyDeviceToHost, s));
        GL_HIP(hipMemcpyAsync(csc_data, d_vals.p, nnz * 4u, hipMemcpyDeviceToHost, s));
        GL_HIP(hipStreamSynchronize(s));
    }
    return GL_OK;
}

int fmt_normalize_by_outdegree(uint32_t num_rows, uint32_t num_cols, const uint32_t *h_indptr, const uint32_t *h_indices, float *h_data) {
    hipStream_t s = ctx().stream;
    const uint64_t nnz = h_indptr[num_rows];
    if (!nnz) return GL_OK;
    DevMem d_cols, d_out, d_deg, d_bad;
    int rc;
    if ((rc = d_cols.alloc(nnz * 4u)) != GL_OK || (rc = d_out.alloc(nnz * 4u)) != GL_OK || (rc = d_deg.alloc((size_t)std::max(num_cols, 1u) * 4u)) != GL_OK ||
        (rc = d_bad.alloc(16)) != GL_OK)
        return rc;
    GL_HIP(hipMemsetAsync(d_deg.p, 0, (size_t)num_cols * 4u, s));
    GL_HIP(hipMemsetAsync(d_bad.p, 0, 16, s));
    GL_HIP(hipMemcpyAsync(d_cols.p, h_indices, nnz * 4u, hipMemcpyHostToDevice, s));
    fmt_degree_kernel<<<flat_grid(nnz), kFmtThreads, 0, s>>>(d_cols.as<uint32_t>(), nnz, num_cols, d_deg.as<uint32_t>(), d_bad.as<uint32_t>());
    GL_LAUNCH_CHECK();
    uint32_t bad = 0;
    GL_HIP(d2h_word_sync(&bad, d_bad.p, s));
    if (bad) return set_error(GL_ERR_INVALID_ARG, "gl_csr_normalize_by_outdegree: column index out of range (num_cols %u)", num_cols);
    fmt_normalize_kernel<<<flat_grid(nnz), kFmtThreads, 0, s>>>(d_cols.as<uint32_t>(), d_deg.as<uint32_t>(), nnz, d_out.as<float>());
    GL_LAUNCH_CHECK();
    GL_HIP(hipMemcpyAsync(h_data, d_out.p, nnz * 4u, hipMemcpyDeviceToHost, s));
    GL_HIP(hipStreamSynchronize(s));
    return GL_OK;
}

int fmt_emit_bool(DevCsr *c, const EmitBool &e, gl_spmv_plan p, uint32_t *max_rows) {
    const uint32_t cb = bits_for(std::max<uint64_t>(e.num_cols, 2u));
    const uint32_t bb = bits_for(std::max<uint32_t>(e.bp->nblocks, 2u));
    if (c->nnz >= 0x7fffffffull) return set_error(GL_ERR_UNSUPPORTED, "plan creation on the device: more than 2^31 - 1 entries in a shard");
    if (bb + cb + 1u <= 32u) return emit_bool_typed<uint32_t>(c, e, p, max_rows, cb, bb);
    return emit_bool_typed<unsigned long long>(c, e, p, max_rows, cb, bb);
}

}  // namespace gl

extern "C" {

int gl_csr2csc(uint32_t num_rows, uint32_t num_cols, const uint32_t *indptr, const uint32_t *indices, const float *data,
               uint32_t *csc_indptr, uint32_t *csc_indices, float *csc_data) {
    GL_ARG(indptr != nullptr && csc_indptr != nullptr);
    const uint64_t nnz = indptr[num_rows];
    GL_ARG(nnz == 0 || (indices != nullptr && data != nullptr && csc_indices != nullptr && csc_data != nullptr));
    if (gl::ctx().initialized && gl::format_on_device(0u, nnz))
        return gl::fmt_csr2csc(num_rows, num_cols, indptr, indices, data, csc_indptr, csc_indices, csc_data);
    return gl::host_csr2csc(num_rows, num_cols, indptr, indices, data, csc_indptr, csc_indices, csc_data);
}

int gl_csr_normalize_by_outdegree(uint32_t num_rows, uint32_t num_cols, const uint32_t *indptr, const uint32_t *indices, float *data) {
    GL_ARG(indptr != nullptr);
    const uint64_t nnz = indptr[num_rows];
    GL_ARG(nnz == 0 || (indices != nullptr && data != nullptr));
    if (gl::ctx().initialized && gl::format_on_device(0u, nnz))
        return gl::fmt_normalize_by_outdegree(num_rows, num_cols, indptr, indices, data);
    std::vector<uint32_t> per_col(num_cols, 0u);
    for (uint64_t i = 0; i < nnz; i++) {
        if (indices[i] >= num_cols)
            return gl::set_error(GL_ERR_INVALID_ARG, "gl_csr_normalize_by_outdegree: column index %u out of range (num_cols %u)", indices[i], num_cols);
        per_col[indices[i]]++;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)nnz; i++) data[i] = (float)(1.0 / (double)per_col[indices[i]]);
    return GL_OK;
}

}  // extern "C"

// gl_init loads this translation unit's code object up front (HIP defers that to the unit's first launch, which would put
// tens of milliseconds into somebody's timed call)
namespace gl {
int preload_format() {
    hipFuncAttributes attr;
    GL_HIP(hipFuncGetAttributes(&attr, (const void *)fmt_degree_kernel));
    return GL_OK;
}
}  // namespace gl
