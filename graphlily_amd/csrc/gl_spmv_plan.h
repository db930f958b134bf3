// Shared by gl_spmv.hip (general semiring SpMV) and gl_spmv_bool.hip ((||,&&)-only SpMV): the plan object,
// the row-block / segment planner and the host-side sorting helpers.
#ifndef GL_SPMV_PLAN_H_
#define GL_SPMV_PLAN_H_

#include "gl_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace gl {

constexpr uint32_t kRowBits = 14;
constexpr uint32_t kRowPad = (1u << kRowBits) - 1u;   // row_in_block value of a padding entry
constexpr uint32_t kMaxBlockRows = kRowPad;            // 16383
constexpr uint32_t kColOffBits = 32 - kRowBits;        // 18
constexpr uint32_t kHubSlots = 16;                     // private accumulators per hub row
constexpr uint32_t kMaxHubRows = 64;                   // per row block
// Padding entries accumulate into DUMMY slots behind a block's last accumulator -- one per lane (a group's padding lanes all
// on one LDS address would serialise the atomic: EXPERIMENTS R3.12), so a block's slots + kPadSlots still fit the 14-bit field
constexpr uint32_t kPadSlots = 64;
constexpr uint32_t kMaxPlainRows = kMaxBlockRows - kHubSlots * kMaxHubRows - kPadSlots;
constexpr uint32_t kBoolMaxPhases = 8;                 // wider matrices keep the general layout
constexpr uint32_t kLdsBudget = 160u * 1024u - 512u;   // per-CU LDS minus a little slack
constexpr uint32_t kThreads = 1024;                    // one workgroup per CU: 16 wavefronts share the tile
constexpr uint32_t kWaves = kThreads / 64;

// (||,&&)-only layout: x as a bitmap, one LDS-resident slice ("phase") of it at a time
constexpr uint32_t kBoolPhaseWords = 36864;                   // 144 KB of x bits
constexpr uint32_t kBoolPhaseCols = kBoolPhaseWords * 32u;    // 1 179 648 columns
#ifndef GL_BOOL_GROUP
#define GL_BOOL_GROUP 256
#endif
constexpr uint32_t kBoolGroup = GL_BOOL_GROUP;                  // entries per group: 16 (256) or 8 (128) bytes per lane and load
constexpr uint32_t kBoolTileWords = (kMaxBlockRows + 1u) / 32u;  // 512: one bit per row slot incl. the padding slot
constexpr uint32_t kBoolHubSlots = 32;                        // private bits per hub row, one per tile word 480..511
constexpr uint32_t kBoolHubMax = 31;                          // hub rows per block (bit 31 of word 511 is the padding slot)
constexpr uint32_t kBoolHubBit0 = (kBoolTileWords - kBoolHubSlots) * 32u;   // 15360: plain rows use the bits below

// Run-coded hot stream (round 5).  A unit's hot entries -- those whose column has a slot in the LDS table -- are sorted by
// slot, so a column's entries form a RUN: the entry keeps a 16-bit row slot (+ its value in the general layout) and the
// column travels as one bit per entry ("the next entry starts a new run") + one base per 64-entry group.  The slots are
// numbered per UNIT (rank among the hot columns that occur in the unit, listed in the unit's `present` array), so that runs
// advance the slot by exactly one and lane l of a group reads table[base + popcount(mask below l)] (v_mbcnt with a scalar mask).
// An ELEMENT is what one wavefront step loads: HG = 4 groups, lane-interleaved -- 8 B of row slots + 16 B of values per lane --
// with a 12-byte header per group kept apart (scalar loads): 6.19 instead of 8 bytes per hot entry.  GENERAL layout only since
// round 6 (the pattern layout, which this coding took from 4 to 2.19 bytes per hot entry in round 5, carries the row-packed
// stream below: it was bound by LDS atomics, not bytes).
// Delta-coded cold stream (round 5).  A unit's cold entries are sorted by gather index, and in the classes the packed vector
// lists first consecutive entries are a few columns apart: the entry keeps a 16-bit row slot (+ its value) and an 8-BIT DELTA to
// its predecessor; entry 0 of a 64-entry group has delta 0 and the group's base (a scalar) is its index.  Lane l's index is
// base + the inclusive prefix sum of the deltas over the lanes -- six DPP adds per PAIR of groups (two 16-bit fields per
// register: 63 x 255 < 2^16).  A gap of more than 255 columns is bridged by DUMMY entries (a dummy slot, value 0, delta 255):
// rare where the vector is dense, one or two per entry only among the rarest columns (a few per cent of the entries).  An
// element = the groups one wavefront step loads, lane-interleaved -- general: 8 B of slots + 4 B of deltas + 16 B of values per
// lane (4 groups, 7 B per entry instead of 8), pattern: 16 B of slots + 8 B of deltas (8 groups, 3 B per entry instead of 4).
constexpr uint32_t kColdGroupsGeneral = 4, kColdGroupsPattern = 8;
constexpr uint32_t kColdElemBytesGeneral = kColdGroupsGeneral * 64u * 7u;   // 512 B of slots, 256 B of deltas, 1024 B of values
constexpr uint32_t kColdElemBytesPattern = kColdGroupsPattern * 64u * 3u;   // 1024 B of slots, 512 B of deltas
constexpr uint32_t kColdMaxDelta = 255;

// Row-packed hot stream (round 6, PATTERN layout only).  rocprofv3's counters put the pattern kernel's LDS at 75 % busy: an f64
// atomic of 64 lanes on random rows costs 24.6 LDS clocks (8 conflict-free: 16 lanes per pass over 16 double-width banks, 2 clocks
// per access), a random 4-byte read 7.1 (scripts/ubench_lds_atomic.hip, profiles/r06_ubench_lds_atomic.txt) -- and a hot entry,
// whose x comes from the LDS table, needs no column locality at all.  So the pattern layout's hot entries are stored ROW-major:
// a RECORD = 7 table slots of ONE row + the row's slot, 8 x 16 bits = 16 bytes, one record per lane and element; a lane looks its
// seven values up, combines them in registers and issues ONE accumulate for the row -- 1/7 of the atomics, no run masks, no
// headers, no per-unit slot numbering (table slots are the plan's global hot slots; the table's slot nhot holds the semiring's
// identity: padding fields of a row's last record name it).  A unit's records are listed by (row, CSR order) and dealt to the lanes
// in 64 contiguous CHUNKS -- element e, lane l = record l * chunk + e -- so that the lanes of one step hold rows that lie far apart
// (a row with more records than a chunk is a hub row, spread over its private slots by lane).  2.29 B per entry + the padding
// (~3 fields per row and block) instead of 2.19.  The general layout keeps the run-coded stream: it is bound by its HBM traffic.
constexpr uint32_t kHotRecEntries = 7;                     // table slots per record
constexpr uint32_t kHotElemBytesRows = 64u * 16u;          // one 16-byte record per lane

constexpr uint32_t kHotGroupsGeneral = 4;
constexpr uint32_t kHotElemBytesGeneral = kHotGroupsGeneral * 64u * 6u;   // 512 B of row slots, then 1024 B of values
constexpr uint32_t kHotHdrWordsPerGroup = 3;                              // per element: HG x {mask lo, mask hi}, then HG bases

struct Shape {
    uint32_t blocks, segments;
};

struct Rec {
    uint32_t col, row_local, val;
};

// stable LSD radix sort of a block's records by column
static inline void sort_by_col(std::vector<Rec> &a, std::vector<Rec> &tmp, uint32_t num_cols) {
    const size_t n = a.size();
    tmp.resize(n);
    int bits = 1;
    while ((1ull << bits) < num_cols) bits++;
    const int passes = (bits + 10) / 11;
    for (int p = 0; p < passes; p++) {
        const int sh = 11 * p;
        size_t cnt[2049] = {0};
        for (size_t i = 0; i < n; i++) cnt[((a[i].col >> sh) & 2047u) + 1]++;
        for (int i = 0; i < 2048; i++) cnt[i + 1] += cnt[i];
        for (size_t i = 0; i < n; i++) tmp[cnt[(a[i].col >> sh) & 2047u]++] = a[i];
        a.swap(tmp);
    }
}

// Row blocks (nnz-balanced boundaries, at most max_rows rows each) and the number of column segments
// ("units", one workgroup each) every block is cut into.  Units are numbered segment-major: all blocks'
// piece 0, then every block's piece 1 (where it exists), ... so that concurrently running workgroups sweep
// the same column window of x.
struct BlockPlan {
    std::vector<uint32_t> bstart;                  // nblocks + 1 row boundaries
    std::vector<uint32_t> seg;                     // segments per block
    std::vector<std::vector<uint32_t>> unit_of;    // [segment][block] -> unit index (0xffffffff: none)
    uint32_t nblocks = 0, nunits = 0, Smax = 1;
    bool all_direct = true;
};

BlockPlan plan_blocks(Shape shape, const uint32_t *h_indptr, uint32_t row_begin, uint32_t row_end, uint32_t max_rows,
                      uint32_t align = 1);

// Where every unit's pieces start, from the cold / hot entry counts of the row blocks (shared by the host and the device
// formatter, which must agree byte for byte): cold groups are budgeted (`dummy_max` bounds the dummy entries a unit can need:
// each advances the gather index by 255, and a unit's indices span at most the gather vector), hot elements and present lists
// are exact / capped by the table size.
struct UnitLayout {
    std::vector<uint64_t> cold_goff;      // nunits + 1: first cold group (multiples of 8)
    std::vector<uint64_t> hot_e0;         // nunits + 1: first hot element
    std::vector<uint64_t> present_off;    // nunits + 1: first entry of the unit's present list (16-bit entries, even offsets)
};
// (mrec != nullptr: the row-packed hot stream -- mrec[b] = records of block b, cut into the block's units by position; an element
//  holds 64 records, no present lists)
inline UnitLayout layout_units(const BlockPlan &bp, const std::vector<uint64_t> &mc, const std::vector<uint64_t> &mh, uint32_t dummy_max,
                               uint32_t hot_groups, uint32_t nhot_table, const std::vector<uint64_t> *mrec = nullptr) {
    UnitLayout ul;
    ul.cold_goff.assign((size_t)bp.nunits + 1, 0);
    ul.hot_e0.assign((size_t)bp.nunits + 1, 0);
    ul.present_off.assign((size_t)bp.nunits + 1, 0);
    for (uint32_t b = 0; b < bp.nblocks; b++) {
        const uint32_t S = bp.seg[b];
        for (uint32_t s = 0; s < S; s++) {
            const size_t u = bp.unit_of[s][b];
            const uint64_t c = mc[b] * (s + 1) / S - mc[b] * s / S, h = mh[b] * (s + 1) / S - mh[b] * s / S;
            ul.cold_goff[u + 1] = c ? ((c + dummy_max + 63) / 64 + 7u) / 8u * 8u : 0u;
            if (mrec) {
                const uint64_t rc = (*mrec)[b] * (s + 1) / S - (*mrec)[b] * s / S;
                ul.hot_e0[u + 1] = (rc + 63) / 64;
                ul.present_off[u + 1] = 0;
            } else {
                ul.hot_e0[u + 1] = ((h + 63) / 64 + hot_groups - 1u) / hot_groups;
                ul.present_off[u + 1] = (std::min<uint64_t>(h, nhot_table) + 1u) / 2u * 2u;
            }
        }
    }
    for (size_t u = 0; u < bp.nunits; u++) {
        ul.cold_goff[u + 1] += ul.cold_goff[u];
        ul.hot_e0[u + 1] += ul.hot_e0[u];
        ul.present_off[u + 1] += ul.present_off[u];
    }
    return ul;
}

}  // namespace gl

struct gl_spmv_plan_s {
    uint32_t num_rows = 0, num_cols = 0, row_begin = 0, row_end = 0;
    uint64_t nnz = 0;
    uint32_t nunits = 0, nblocks = 0, segments = 1, max_block_rows = 0;
    uint64_t ngroups = 0;
    uint2 *d_entries = nullptr;      // delta-coded cold elements (gl::kColdElemBytes*), addressed as bytes
    uint32_t *d_bases = nullptr;     // one base gather index per cold group
    uint4 *d_units = nullptr;        // 3 per unit
    unsigned char *d_hot = nullptr;  // run-coded hot elements (gl::kHotElemBytes*)
    uint32_t *d_hot_hdr = nullptr;   // their headers
    uint16_t *d_present = nullptr;   // per unit: the hot-table slots that occur in it, ascending
    uint32_t nhot_lds = 0;           // longest present list, rounded up to 64: the LDS table's length
    uint64_t nhot_elems = 0;
    uint32_t *d_hub_rows = nullptr;
    uint32_t flags = 0;        // GL_PLAN_* given at creation
    uint32_t nhot = 0;         // cached ("hot") columns, multiple of 64
    uint64_t hot_nnz = 0;      // non-zeros served from the LDS table
    int mix = 0;               // cold/hot groups per iteration: 0 = (4,0) no hot table, 5 = (3,3) default; others for tuning
    uint32_t *d_hot_cols = nullptr;
    float *d_hot_x = nullptr;
    float *d_hot_colval = nullptr;   // pattern plans: the hot columns' values
    // compact gather vector: when a tenth or more of the columns are never gathered (no entry in this shard, or hot),
    // the cold entries index a packed copy of x (pattern plans: of z) that a per-run kernel fills -- fewer lines
    // per sweep of a row block
    uint32_t ncompact = 0;           // 0: gather from x (z) directly
    uint32_t *d_ccols = nullptr;     // packed index -> column, ascending
    float *d_xc = nullptr;           // general plans: x[ccols[j]]
    // CHAINED runs (gl_spmv_plan_chain; round 6): an iterative caller feeds every result straight back as the next x (PageRank, SSSP
    // pull).  The epilogue of such a run also stores y in the next run's packed form -- its slot of the packed vector / hot table,
    // times the column's value in pattern plans: what the helper launch would do -- into the TWIN of the buffers the run reads, and
    // the next run skips the helper.  chain_ptr: the y whose packed form the buffers of `chain_sel` hold (null: none).
    float *d_packed_twin = nullptr, *d_hot_x_twin = nullptr;
    size_t packed_len = 0, hot_x_len = 0;   // floats in d_z / d_xc and in d_hot_x
    bool chain_on = false;
    const float *chain_ptr = nullptr;
    int chain_sel = 0, chain_op = -1;
    uint32_t *d_colmap = nullptr;    // non-null: the vectors are refilled by one streaming pass over x (spmv_spread_x_kernel):
                                     // per column 0x80000000 | hot slot, packed index, or 0xffffffff
    float *d_colval_bycol = nullptr; // pattern plans in that mode: the column values indexed by column
    bool self_hot = false;           // no helper launch: the workgroups gather their (small) hot table from x themselves
    bool wide = false;               // general layout with lane-interleaved group pairs (16-byte stream loads)
    bool pattern = false;            // every column's values are equal: 4-byte entries, z = colval (x) x per run
    float *d_colval = nullptr, *d_z = nullptr;
    float *d_diag = nullptr;         // pattern plans whose diagonal differs from the column values: A[r][r] per local row
    uint32_t *d_diag_has = nullptr;
    uint4 *d_blocks = nullptr;       // {first row, #rows, #segments, -} per row block
    float *d_partials = nullptr;     // split plans: segments x rows planes of per-unit tiles
    uint32_t max_plain_rows = 0;     // tallest block without hub slots
    // (||,&&)-only layout (GL_PLAN_BOOLEAN): 4-byte pattern entries, x packed to bits once per run
    bool boolean = false;
    int bool_compressed = 0;       // the groups are delta-coded, 3 bytes per entry (bool_plan_compress): 1 = 8-bit deltas, 2 = 10-bit
    uint32_t nphases = 0;
    uint4 *d_spans = nullptr;      // {first word of the phase in xbits, first group, end group, lo4 | hi4 << 16}
    uint32_t *d_xbits = nullptr;   // nphases * kBoolPhaseWords words
    // whole-matrix boolean plans also keep the rows as plain CSR (4 B per non-zero more): gl_bfs_bits_push_step's bottom-up
    // branch scans the rows a BFS has not reached yet; zero-valued entries carry the column 0xffffffff
    // (row shards keep theirs too: d_csr_indptr[r - row_begin] are offsets into the WHOLE matrix' entry list, of which
    // d_csr_indices holds the shard's part starting at csr_nz_base)
    uint32_t *d_csr_indptr = nullptr, *d_csr_indices = nullptr;
    uint32_t csr_nz_base = 0;
    // GL_PLAN_REFERENCE_ORDER: the shard's plain CSR (indptr rebased to 0, values kept), evaluated a thread per row in
    // the reference's own order -- a diagnostic layout, not a fast one
    bool reference_order = false;
    float *d_csr_data = nullptr;
    uint64_t device_bytes = 0;
    size_t b_entries = 0, b_bases = 0, b_units = 0, b_hub_rows = 0, b_spans = 0;   // sizes of the formatted arrays (gl_spmv_plan_export)
    size_t b_hot = 0, b_hot_hdr = 0, b_present = 0;
};

namespace gl {
// ------------------------------------------------------------------------------------------ gl_format.hip
// Plan creation on the GPU.  The O(rows + columns) decisions of a plan -- row blocks, hot columns,
// packed gather order, hub rows, group budgets -- stay on the host and are shared with the host formatter; every
// O(nnz) step runs on the device over a staged copy of the shard's CSR: column degrees, column-constant ("pattern")
// detection, and the per-block column sort + group packing that produces the entry stream.  Both formatters
// produce byte-identical device arrays (tests/test_gpu_format.py).
struct DevCsr;   // the shard's indptr / indices / data on the device
int devcsr_stage(DevCsr **out, const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data,
                 uint32_t row_begin, uint32_t row_end);
void devcsr_release(DevCsr *c);
// hand the staged rows over to a plan (the bottom-up BFS step reads them); zero-valued entries get column 0xffffffff
int devcsr_adopt_rows(DevCsr *c, uint32_t **d_indptr, uint32_t **d_indices);
bool format_on_device(uint32_t flags, uint64_t nnz);   // policy: GL_PLAN_HOST_FORMAT / GRAPHLILY_PLAN_DEVICE / size
int fmt_column_degrees(DevCsr *c, uint32_t num_cols, std::vector<uint32_t> &deg, int *bad_col);
int fmt_detect_pattern(DevCsr *c, uint32_t num_cols, std::vector<uint32_t> &colbits, std::vector<uint32_t> &diag_has,
                       std::vector<float> &diag_val, int *mismatch, uint64_t *exceptions);

struct EmitGeneral {   // what the host planner decided (gl_spmv_plan_create_ex)
    const BlockPlan *bp;
    uint32_t dummy_max;                // layout_units' bound on the dummy entries of a unit
    const uint32_t *colmap;            // per column: 0x80000000 | hot slot, or the index the cold entry gathers from
    uint32_t gather_cols, nhot_table;  // ranges of those two index spaces
    bool diag_mode;                    // diagonal entries that differ from their column's value are dropped
    const uint32_t *colbits;           // host copy of the column values (diag_mode)
    const uint32_t *diag_has;          // host bitmap of the rows whose diagonal entry is an exception (diag_mode)
    bool pattern, wide;                // (pattern plans carry the ROW-PACKED hot stream)
    uint32_t group_mult;
    uint32_t hub_div;
    const uint32_t *h_indptr;          // host indptr (global), for the per-row counts
    uint32_t num_cols;
};
// fills p->d_entries / d_bases / d_units / d_hub_rows / d_hot / d_hot_hdr / d_present (allocated here), p->ngroups,
// p->nhot_elems, p->nhot_lds and hub_count, hot_nnz
int fmt_emit_general(DevCsr *c, const EmitGeneral &e, gl_spmv_plan p, std::vector<uint32_t> &hub_count, uint64_t *hot_nnz);
// (||,&&) layout: the device twin of bool_plan_build's record loop (gl_spmv_bool.hip)
struct EmitBool {
    const BlockPlan *bp;
    const uint32_t *h_indptr;
    uint32_t num_cols;
};
int fmt_emit_bool(DevCsr *c, const EmitBool &e, gl_spmv_plan p, uint32_t *max_rows);

// SpMSpV plans: the {row, value} stream and its column pointers for the rows of [r0, r1), built on the device from a
// host CSC (replaces a serial host loop); indptr_out is the host copy of the new column pointers
int fmt_spmspv_stream(uint32_t num_rows, uint32_t num_cols, const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data,
                      uint32_t r0, uint32_t r1, uint32_t **d_indptr_out, uint2 **d_stream_out, std::vector<uint32_t> &indptr_out);
int fmt_csr2csc(uint32_t num_rows, uint32_t num_cols, const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data,
                uint32_t *csc_indptr, uint32_t *csc_indices, float *csc_data);
int fmt_normalize_by_outdegree(uint32_t num_rows, uint32_t num_cols, const uint32_t *h_indptr, const uint32_t *h_indices, float *h_data);

// gl_spmv_bool.hip
int bool_plan_build(gl_spmv_plan p, const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data);
int bool_plan_compress(gl_spmv_plan p);   // after the build: 4-byte entries -> 3 (16-bit row slot + 8-bit column delta) where every group allows it
int bool_plan_run(gl_spmv_plan p, const float *d_x, const uint32_t *bits, const float *d_mask, float *d_y, float zero,
                  int mask_type, hipStream_t s);
int pack_bits(const float *d_x, uint32_t n, uint32_t *d_bits, hipStream_t s);
int unpack_bits(const uint32_t *d_bits, uint32_t n, float *d_x, hipStream_t s);
int bfs_bits_begin_from(uint32_t *d_ctl, uint32_t ctl_words, const float *d_x, uint32_t n, uint32_t *d_bits, uint32_t words, hipStream_t s,
                        const float *d_distance, gl_spmv_plan rows);
int bool_plan_bfs_step(gl_spmv_plan p, const uint32_t *bits_in, uint32_t *bits_out, float *d_distance, float level, hipStream_t s,
                       const uint32_t *gate = nullptr, uint32_t gate_value = 0, int gate_op = GL_GATE_EQ, uint32_t *back_ctl = nullptr,
                       uint32_t back_slot = 0, float back_threshold = 0.0f, int back_may_continue = 0, const BfsBitsCtl *v2 = nullptr,
                       const uint32_t *v2_indptr = nullptr, uint32_t v2_ncols = 0, uint32_t *v2_push_acc = nullptr, bool v2_deferred = false);
int bool_plan_run_bits(gl_spmv_plan p, float *d_y, const uint32_t *run_flag, hipStream_t s, const uint32_t *xbits = nullptr);
// gl_bfs_shard.h: one slot of the bit-frontier BFS schedule on a row shard in one launch
struct BfsPushArgs;
struct BfsShardArgs;
int bool_plan_bfs_shard_step(gl_spmv_plan p, BfsPushArgs pa, BfsShardArgs sa, hipStream_t s);
// gl_apply.hip: the set bits of d_bits[0..n) as a sparse list {row, 1} with head {count, 0}; no-op unless *gate_word == gate_value
int bits_to_sparse_gated(const uint32_t *d_bits, uint32_t n, gl_idx_val *d_out, uint32_t *d_counts, const uint32_t *gate_word,
                         uint32_t gate_value, hipStream_t s);
uint32_t *bool_plan_xbits(gl_spmv_plan p);
size_t bool_plan_xbits_bytes(gl_spmv_plan p);
int spmv_run_general(gl_spmv_plan p, const float *d_x, const float *d_mask, float *d_y, int op, float zero, int mask_type,
                     const uint32_t *run_flag);
// gl_spmspv.hip: what the pull step of the bit-frontier BFS schedule (gl_bfs_bits_pull_step) needs from the CSC plan
unsigned long long spmspv_heavy_work(gl_spmspv_plan p);
unsigned long long spmspv_bottom_up_limit(gl_spmspv_plan p);
unsigned long long spmspv_plan_nnz(gl_spmspv_plan p);
const void *spmspv_plan_bfs_rows(gl_spmspv_plan p);   // the plan the slot's push step (enqueued first) can scan bottom-up
const uint32_t *spmspv_plan_indptr(gl_spmspv_plan p);
uint32_t spmspv_plan_num_cols(gl_spmspv_plan p);
uint32_t *spmspv_plan_bfs_acc(gl_spmspv_plan p);   // 64 lines of 32 words: {new vertices, -, column lengths (64 bits)} of a push step
bool spmspv_plan_whole(gl_spmspv_plan p, uint32_t num_rows);
// gl_spmspv.hip: forget `dying` wherever gl_spmspv_plan_attach_pull attached it
void spmspv_detach_everywhere(gl_spmv_plan dying);
// gl_spmv.hip: y initialisation for plans whose units fold into y
int spmv_init_rows(int op, int mask_type, uint32_t r0, uint32_t r1, const float *mask, float *y, float zero, hipStream_t s);
}  // namespace gl

#endif  // GL_SPMV_PLAN_H_
