// SpMV over a semiring for MI355X (gfx950): y = mask( zero (+) A (x) x ).
//
// Replaces the FPGA path  SpMVModule::load_and_format_matrix -> csr2cpsr
// (module/spmv_module.h:281-370, io/data_formatter.h:456-534) and
// kernel_spmv (hw/kernel_spmv_impl.h:392-819).
//
// Layout ("row-segment stream", the CDNA4 counterpart of the FPGA's cyclic
// packed streams with in-band end-of-row markers, io/data_formatter.h:54-81):
//   stream[k] = { col | row_end << 31 , val }   8 bytes per non-zero, CSR order,
//               so a wavefront step reads 64 x 8 = 512 contiguous bytes;
//   tiles[t]  = { stream offset, first row, nnz | flags << 16, aux }
//               one wavefront per tile; tiles are cut at row boundaries
//               (<= tile_nnz non-zeros) so no partial sums cross tiles, except
//               for rows longer than a tile, which become runs of LONG tiles
//               whose partials are combined by a second, tiny kernel;
//   empty rows are listed separately and written by trailing blocks of the
//   same launch (the reference's skip_empty_rows idea, spmv_module.h:199).
// Per step a lane gathers x[col] through L1/L2, multiplies, and the wave does
// a segmented scan keyed by the ballot of row_end flags; lanes holding a
// row_end write y (mask fused into the epilogue like write_to_out_ddr,
// hw/kernel_spmv_impl.h:339-389).
#include "gl_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace gl {

constexpr uint32_t kTileLong = 1u;       // tile is a slice of one long row (no row_end flags inside)
constexpr uint32_t kTileNonContig = 2u;  // rows of the tile are not consecutive: go through rowmap[aux + k]

struct SpmvArgs {
    const uint2 *stream;
    const uint4 *tiles;
    const uint32_t *rowmap;
    const uint32_t *empty_rows;
    const float *x;
    const float *mask;
    float *y;
    float *long_partials;
    uint32_t ntiles;
    uint32_t tile_blocks;
    uint32_t nempty;
    float zero;
};

template <int OP, int MASK>
__device__ __forceinline__ void store_row(const SpmvArgs &a, uint32_t row, float acc) {
    using S = Semiring<OP>;
    float out = S::finish(a.zero, acc);
    if (MASK != GL_NOMASK) {
        // masked-off rows are literal 0, and the mask is compared with 0 (spmv_module.h:518-530)
        if (!mask_allows<MASK>(a.mask[row], 0.0f)) out = 0.0f;
    }
    a.y[row] = out;
}

template <int OP, int MASK, int U>
__global__ __launch_bounds__(256) void spmv_rseg_kernel(SpmvArgs a) {
    using S = Semiring<OP>;
    const uint32_t lane = threadIdx.x & 63u;

    if (blockIdx.x >= a.tile_blocks) {
        // rows without any non-zero: y = semiring zero (accumulator never touched)
        uint32_t i = (blockIdx.x - a.tile_blocks) * 256u + threadIdx.x;
        if (i < a.nempty) {
            uint32_t r = a.empty_rows[i];
            float out = a.zero;
            if (MASK != GL_NOMASK) {
                if (!mask_allows<MASK>(a.mask[r], 0.0f)) out = 0.0f;
            }
            a.y[r] = out;
        }
        return;
    }

    const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (t >= a.ntiles) return;
    const uint4 d = a.tiles[t];
    const uint32_t off = __builtin_amdgcn_readfirstlane(d.x);
    const uint32_t row0 = __builtin_amdgcn_readfirstlane(d.y);
    const uint32_t cnt = __builtin_amdgcn_readfirstlane(d.z) & 0xffffu;
    const uint32_t flags = __builtin_amdgcn_readfirstlane(d.z) >> 16;
    const uint32_t aux = __builtin_amdgcn_readfirstlane(d.w);
    const uint2 *__restrict__ sp = a.stream + off;
    const float ident = S::ident(a.zero);

    // Software pipeline, three stages deep, so that a wave always has stream loads AND gathers in
    // flight while it reduces:   stream(i+2)  |  gather x(i+1)  |  reduce(i).
    // c0/x0 = entries and gathered x of the iteration being reduced, c1 = entries of the next one.
    const uint32_t kStep = 64u * U;
    uint2 c0[U], c1[U];
    float x0[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        uint32_t k = u * 64u + lane;
        c0[u] = (k < cnt) ? load_stream_nt(sp + k) : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        uint32_t k = kStep + u * 64u + lane;
        c1[u] = (k < cnt) ? load_stream_nt(sp + k) : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < U; u++) x0[u] = a.x[c0[u].x & 0x7fffffffu];

    if (flags & kTileLong) {
        // slice of one long row: plain per-lane accumulation, one wave reduction at the end
        float acc = ident;
        for (uint32_t base = 0; base < cnt; base += kStep) {
            uint2 c2[U];
            float x1[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                uint32_t k = base + 2u * kStep + u * 64u + lane;
                c2[u] = (k < cnt) ? load_stream_nt(sp + k) : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < U; u++) x1[u] = a.x[c1[u].x & 0x7fffffffu];
#pragma unroll
            for (int u = 0; u < U; u++) {
                uint32_t k = base + u * 64u + lane;
                float p = (k < cnt) ? S::mul(__uint_as_float(c0[u].y), x0[u]) : ident;
                acc = S::add(acc, p);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                c0[u] = c1[u];
                c1[u] = c2[u];
                x0[u] = x1[u];
            }
        }
#pragma unroll
        for (int dlt = 32; dlt >= 1; dlt >>= 1) acc = S::add(acc, __shfl_down(acc, dlt));
        if (lane == 0) a.long_partials[aux] = acc;
        return;
    }

    float carry = ident;     // partial sum of the row that is open at the start of the step
    uint32_t rows_done = 0;  // rows of this tile already written
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    for (uint32_t base = 0; base < cnt; base += kStep) {
        uint2 c2[U];
        float x1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t k = base + 2u * kStep + u * 64u + lane;
            c2[u] = (k < cnt) ? load_stream_nt(sp + k) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < U; u++) x1[u] = a.x[c1[u].x & 0x7fffffffu];

#pragma unroll
        for (int u = 0; u < U; u++) {
            if (base + u * 64u >= cnt) break;  // wave-uniform
            const uint32_t k = base + u * 64u + lane;
            const bool valid = k < cnt;
            float p = valid ? S::mul(__uint_as_float(c0[u].y), x0[u]) : ident;
            const bool e = valid && (c0[u].x >> 31);
            const uint64_t me = __ballot(e);
            const uint64_t below = me & lt_mask;
            // first lane of the segment this lane belongs to
            const uint32_t seg_start = below ? (64u - (uint32_t)__clzll((long long)below)) : 0u;
#pragma unroll
            for (uint32_t dlt = 1; dlt < 64; dlt <<= 1) {
                float up = __shfl_up(p, dlt);
                if (lane >= seg_start + dlt) p = S::add(p, up);
            }
            const float v = (seg_start == 0u) ? S::add(carry, p) : p;
            if (e) {
                uint32_t kr = rows_done + (uint32_t)__popcll(below);
                uint32_t row = (flags & kTileNonContig) ? a.rowmap[aux + kr] : row0 + kr;
                store_row<OP, MASK>(a, row, v);
            }
            rows_done += (uint32_t)__popcll(me);
            const float last = __shfl(v, 63);
            carry = (me >> 63) ? ident : last;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            c0[u] = c1[u];
            c1[u] = c2[u];
            x0[u] = x1[u];
        }
    }
}

struct LongRowArgs {
    const uint4 *long_rows;  // {row, first partial, nparts, 0}
    const float *partials;
    const float *mask;
    float *y;
    uint32_t nlong;
    float zero;
};

// one wavefront per long row; fixed combination order => deterministic result
template <int OP, int MASK>
__global__ __launch_bounds__(256) void spmv_long_rows_kernel(LongRowArgs a) {
    using S = Semiring<OP>;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (i >= a.nlong) return;
    const uint4 d = a.long_rows[i];
    float acc = S::ident(a.zero);
    for (uint32_t k = lane; k < d.z; k += 64u) acc = S::add(acc, a.partials[d.y + k]);
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1) acc = S::add(acc, __shfl_down(acc, dlt));
    if (lane == 0) {
        float out = S::finish(a.zero, acc);
        if (MASK != GL_NOMASK) {
            if (!mask_allows<MASK>(a.mask[d.x], 0.0f)) out = 0.0f;
        }
        a.y[d.x] = out;
    }
}

static int unroll_setting() {
    static int u = [] {
        const char *e = getenv("GRAPHLILY_SPMV_UNROLL");
        int v = e ? atoi(e) : 4;
        return (v == 1 || v == 2 || v == 4 || v == 8) ? v : 4;
    }();
    return u;
}

template <int OP, int MASK>
static int launch_spmv(const SpmvArgs &a, const LongRowArgs &l, hipStream_t s) {
    unsigned blocks = a.tile_blocks + cdiv(a.nempty, 256);
    if (blocks) {
        Profiler &pf = prof();
        const bool timed = pf.on && 2ull * (pf.used + 1) <= pf.events.size();
        if (timed) GL_HIP(hipEventRecord(pf.events[2 * pf.used], s));
        switch (unroll_setting()) {
            case 1: spmv_rseg_kernel<OP, MASK, 1><<<blocks, 256, 0, s>>>(a); break;
            case 2: spmv_rseg_kernel<OP, MASK, 2><<<blocks, 256, 0, s>>>(a); break;
            case 8: spmv_rseg_kernel<OP, MASK, 8><<<blocks, 256, 0, s>>>(a); break;
            default: spmv_rseg_kernel<OP, MASK, 4><<<blocks, 256, 0, s>>>(a); break;
        }
        GL_LAUNCH_CHECK();
        if (timed) {
            GL_HIP(hipEventRecord(pf.events[2 * pf.used + 1], s));
            pf.used++;
        }
    }
    if (l.nlong) {
        spmv_long_rows_kernel<OP, MASK><<<cdiv(l.nlong, 4), 256, 0, s>>>(l);
        GL_LAUNCH_CHECK();
    }
    return GL_OK;
}

template <int OP>
static int dispatch_mask(int mask_type, const SpmvArgs &a, const LongRowArgs &l, hipStream_t s) {
    switch (mask_type) {
        case GL_NOMASK: return launch_spmv<OP, GL_NOMASK>(a, l, s);
        case GL_MASK_WRITETOZERO: return launch_spmv<OP, GL_MASK_WRITETOZERO>(a, l, s);
        case GL_MASK_WRITETOONE: return launch_spmv<OP, GL_MASK_WRITETOONE>(a, l, s);
        default: return set_error(GL_ERR_INVALID_ARG, "gl_spmv_run: invalid mask type %d", mask_type);
    }
}

}  // namespace gl

struct gl_spmv_plan_s {
    uint32_t num_rows = 0, num_cols = 0, row_begin = 0, row_end = 0;
    uint64_t nnz = 0;
    uint32_t ntiles = 0, nempty = 0, nlong = 0, nparts = 0;
    uint2 *d_stream = nullptr;
    uint4 *d_tiles = nullptr;
    uint32_t *d_rowmap = nullptr;
    uint32_t *d_empty = nullptr;
    uint4 *d_long_rows = nullptr;
    float *d_long_partials = nullptr;
    uint64_t device_bytes = 0;
};

namespace gl {

static uint32_t tile_nnz_setting() {
    // tuning knob; the reference passes (out_buf_len, vec_buf_len) hints for the same purpose
    const char *e = getenv("GRAPHLILY_SPMV_TILE_NNZ");
    long v = e ? atol(e) : 2048;
    if (v < 64) v = 64;
    if (v > 32768) v = 32768;
    return (uint32_t)v;
}

template <typename T>
static int upload(T **d, const std::vector<T> &h, uint64_t *bytes) {
    size_t n = h.size() * sizeof(T);
    GL_HIP(hipMalloc((void **)d, n ? n : 16));
    if (n) GL_HIP(hipMemcpy(*d, h.data(), n, hipMemcpyHostToDevice));
    *bytes += n;
    return GL_OK;
}

}  // namespace gl

extern "C" {

int gl_spmv_plan_create(gl_spmv_plan *plan, uint32_t num_rows, uint32_t num_cols,
                        const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data,
                        uint32_t row_begin, uint32_t row_end) {
    GL_REQUIRE_INIT();
    GL_ARG(plan != nullptr && h_indptr != nullptr);
    GL_ARG(row_begin <= row_end && row_end <= num_rows);
    GL_ARG(num_cols < 0x80000000u);
    const uint64_t nz0 = h_indptr[row_begin], nz1 = h_indptr[row_end];
    GL_ARG(nz1 >= nz0);
    const uint64_t nnz = nz1 - nz0;
    GL_ARG(nnz == 0 || (h_indices != nullptr && h_data != nullptr));

    const uint32_t tile_nnz = gl::tile_nnz_setting();
    std::vector<uint4> tiles;
    std::vector<uint32_t> rowmap, empty;
    std::vector<uint4> long_rows;
    std::vector<uint2> stream(nnz);
    tiles.reserve(nnz / tile_nnz * 5 / 4 + 16);
    bool any_noncontig = false;

    // open tile state
    uint32_t cur_cnt = 0, cur_off = 0, cur_row0 = 0, cur_last = 0, cur_map0 = 0;
    bool cur_noncontig = false;
    auto flush = [&]() {
        if (!cur_cnt) return;
        uint32_t flags = cur_noncontig ? gl::kTileNonContig : 0u;
        any_noncontig |= cur_noncontig;
        tiles.push_back(make_uint4(cur_off, cur_row0, cur_cnt | (flags << 16), cur_map0));
        cur_cnt = 0;
        cur_noncontig = false;
    };
    uint32_t nparts = 0;
    for (uint32_t r = row_begin; r < row_end; r++) {
        const uint64_t s = h_indptr[r], e = h_indptr[r + 1];
        GL_ARG(e >= s && e <= nz1);
        const uint64_t len = e - s;
        if (len == 0) {
            empty.push_back(r);
            continue;
        }
        if (len > tile_nnz) {
            flush();
            uint32_t parts = (uint32_t)((len + tile_nnz - 1) / tile_nnz);
            long_rows.push_back(make_uint4(r, nparts, parts, 0u));
            for (uint32_t p = 0; p < parts; p++) {
                uint64_t ps = s + (uint64_t)p * tile_nnz;
                uint32_t pc = (uint32_t)std::min<uint64_t>(tile_nnz, e - ps);
                tiles.push_back(make_uint4((uint32_t)(ps - nz0), r, pc | (gl::kTileLong << 16), nparts + p));
            }
            nparts += parts;
            continue;
        }
        if (cur_cnt && cur_cnt + len > tile_nnz) flush();
        if (!cur_cnt) {
            cur_off = (uint32_t)(s - nz0);
            cur_row0 = r;
            cur_map0 = (uint32_t)rowmap.size();
        } else if (r != cur_last + 1) {
            cur_noncontig = true;
        }
        rowmap.push_back(r);
        cur_last = r;
        cur_cnt += (uint32_t)len;
    }
    flush();

    // fill the stream: CSR order, row_end flag on the last entry of every short row
    {
        std::vector<uint8_t> is_long;  // only consulted when long rows exist
        const bool have_long = !long_rows.empty();
        if (have_long) {
            is_long.assign(row_end - row_begin, 0);
            for (const uint4 &lr : long_rows) is_long[lr.x - row_begin] = 1;
        }
#pragma omp parallel for schedule(static, 4096)
        for (int64_t r = row_begin; r < (int64_t)row_end; r++) {
            const uint64_t s = h_indptr[r], e = h_indptr[r + 1];
            if (s == e) continue;
            for (uint64_t i = s; i < e; i++) stream[i - nz0] = make_uint2(h_indices[i], __builtin_bit_cast(uint32_t, h_data[i]));
            if (!(have_long && is_long[r - row_begin])) stream[e - 1 - nz0].x |= 0x80000000u;
        }
    }
    for (uint64_t i = 0; i < nnz; i++) {
        if ((stream[i].x & 0x7fffffffu) >= num_cols)
            return gl::set_error(GL_ERR_INVALID_ARG, "gl_spmv_plan_create: column index %u out of range (num_cols %u)",
                                 stream[i].x & 0x7fffffffu, num_cols);
    }
    if (!any_noncontig) rowmap.clear();

    gl_spmv_plan p = new gl_spmv_plan_s();
    p->num_rows = num_rows;
    p->num_cols = num_cols;
    p->row_begin = row_begin;
    p->row_end = row_end;
    p->nnz = nnz;
    p->ntiles = (uint32_t)tiles.size();
    p->nempty = (uint32_t)empty.size();
    p->nlong = (uint32_t)long_rows.size();
    p->nparts = nparts;
    int rc;
    if ((rc = gl::upload(&p->d_stream, stream, &p->device_bytes)) != GL_OK ||
        (rc = gl::upload(&p->d_tiles, tiles, &p->device_bytes)) != GL_OK ||
        (rc = gl::upload(&p->d_rowmap, rowmap, &p->device_bytes)) != GL_OK ||
        (rc = gl::upload(&p->d_empty, empty, &p->device_bytes)) != GL_OK ||
        (rc = gl::upload(&p->d_long_rows, long_rows, &p->device_bytes)) != GL_OK) {
        gl_spmv_plan_destroy(p);
        return rc;
    }
    hipError_t he = hipMalloc((void **)&p->d_long_partials, (nparts ? nparts : 4) * sizeof(float));
    if (he != hipSuccess) {
        gl_spmv_plan_destroy(p);
        return gl::set_error(GL_ERR_HIP, "hipMalloc(long partials): %s", hipGetErrorString(he));
    }
    p->device_bytes += (uint64_t)nparts * sizeof(float);
    *plan = p;
    return GL_OK;
}

int gl_spmv_plan_destroy(gl_spmv_plan p) {
    if (!p) return GL_OK;
    (void)hipFree(p->d_stream);
    (void)hipFree(p->d_tiles);
    (void)hipFree(p->d_rowmap);
    (void)hipFree(p->d_empty);
    (void)hipFree(p->d_long_rows);
    (void)hipFree(p->d_long_partials);
    delete p;
    return GL_OK;
}

int gl_spmv_plan_info(gl_spmv_plan p, uint64_t *nnz, uint64_t *device_bytes, uint32_t *num_tiles) {
    GL_ARG(p != nullptr);
    if (nnz) *nnz = p->nnz;
    if (device_bytes) *device_bytes = p->device_bytes;
    if (num_tiles) *num_tiles = p->ntiles;
    return GL_OK;
}

int gl_spmv_run(gl_spmv_plan p, const float *d_x, const float *d_mask, float *d_y, int op, float zero,
                int mask_type) {
    GL_REQUIRE_INIT();
    GL_ARG(p != nullptr && d_y != nullptr);
    GL_ARG(d_x != nullptr || p->nnz == 0);
    GL_ARG(mask_type == GL_NOMASK || d_mask != nullptr);
    gl::SpmvArgs a;
    a.stream = p->d_stream;
    a.tiles = p->d_tiles;
    a.rowmap = p->d_rowmap;
    a.empty_rows = p->d_empty;
    a.x = d_x;
    a.mask = d_mask;
    a.y = d_y;
    a.long_partials = p->d_long_partials;
    a.ntiles = p->ntiles;
    a.tile_blocks = gl::cdiv(p->ntiles, 4);
    a.nempty = p->nempty;
    a.zero = zero;
    gl::LongRowArgs l;
    l.long_rows = p->d_long_rows;
    l.partials = p->d_long_partials;
    l.mask = d_mask;
    l.y = d_y;
    l.nlong = p->nlong;
    l.zero = zero;
    hipStream_t s = gl::ctx().stream;
    switch (op) {
        case GL_OP_MULADD: return gl::dispatch_mask<GL_OP_MULADD>(mask_type, a, l, s);
        case GL_OP_ANDOR: return gl::dispatch_mask<GL_OP_ANDOR>(mask_type, a, l, s);
        case GL_OP_ADDMIN: return gl::dispatch_mask<GL_OP_ADDMIN>(mask_type, a, l, s);
        default: return gl::set_error(GL_ERR_INVALID_ARG, "gl_spmv_run: invalid semiring op %d", op);
    }
}

}  // extern "C"
