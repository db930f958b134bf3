/*
 * graphlily_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of the CPU reference path of
 * cornell-zhang/GraphLily (the `compute_reference_results` members of the
 * graphlily::module classes and the app-level compositions built on them).
 * Every function cites the reference file:line whose behaviour it follows.
 * Nothing in the shipped product (graphlily_amd/, include/) may include, link
 * or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg use it, and only as the checker / the timed CPU baseline.
 *
 * Parity pin (see DESIGN.md "Oracle"): the reference itself cannot be built in
 * this image without stand-ins for ap_fixed.h / cnpy.h / CL/cl_ext_xilinx.h /
 * gtest, so `oracle/_ref` does not exist.  This restatement is pinned against
 *   (1) the known-answer tests of the reference's tests/test_io.cpp:68-140
 *       (loader, csr2csc, dim rounding, out-degree normalisation),
 *   (2) the two data fixtures the reference ships (the two .npz files under tests/test_data),
 *   (3) the semiring x mask and app-level known answers recorded in
 *       SURVEY.md section 8(c), which the survey step captured from the
 *       compiled reference (val_t = float) in this container.
 * Those vectors live under tests/golden/.
 *
 * Arithmetic notes: val_t is float.  All accumulations are sequential in CSR
 * (resp. CSC, frontier) order with a float accumulator, exactly like the
 * reference loops; build with -ffp-contract=off so no FMA contraction changes
 * the (+,x) rounding.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* graphlily/global.h:83-87 */
enum { ORC_MULADD = 0, ORC_ANDOR = 1, ORC_ADDMIN = 2 };
/* graphlily/global.h:103-107 */
enum { ORC_NOMASK = 0, ORC_WRITETOZERO = 1, ORC_WRITETOONE = 2 };
/* graphlily/global.h:80 (float(999999999) == 1e9f) */
#define ORC_FLOAT_INF 999999999.0f

/* graphlily/global.h:69-70: {idx_t index; val_t val;} -- 8 bytes */
typedef struct { uint32_t index; float val; } orc_idx_val_t;

/* ------------------------------------------------------------------ io --- */

/* io/data_loader.h:108-144  csr2csc: counting-sort transpose; rows inside a
 * column come out ascending because rows are visited in order. */
void orc_csr2csc(uint32_t num_rows, uint32_t num_cols,
                 const uint32_t *indptr, const uint32_t *indices, const float *data,
                 uint32_t *csc_indptr, uint32_t *csc_indices, float *csc_data)
{
    uint32_t nnz = indptr[num_rows];
    uint32_t *cnt = (uint32_t *)calloc(num_cols ? num_cols : 1, sizeof(uint32_t));
    for (uint32_t n = 0; n < nnz; n++) cnt[indices[n]]++;
    csc_indptr[0] = 0;
    for (uint32_t c = 0; c < num_cols; c++) csc_indptr[c + 1] = csc_indptr[c] + cnt[c];
    memset(cnt, 0, (size_t)num_cols * sizeof(uint32_t));
    for (uint32_t r = 0; r < num_rows; r++) {
        for (uint32_t i = indptr[r]; i < indptr[r + 1]; i++) {
            uint32_t c = indices[i];
            uint32_t dest = csc_indptr[c] + cnt[c];
            csc_indices[dest] = r;
            csc_data[dest] = data[i];
            cnt[c]++;
        }
    }
    free(cnt);
}

/* io/data_formatter.h:18-33  util_round_csr_matrix_dim: returns the padded
 * dims; the caller extends indptr by repeating indptr[num_rows]. */
void orc_round_dim(uint32_t num_rows, uint32_t num_cols, uint32_t row_div, uint32_t col_div,
                   uint32_t *out_rows, uint32_t *out_cols)
{
    *out_rows = (num_rows % row_div) ? num_rows + (row_div - num_rows % row_div) : num_rows;
    *out_cols = (num_cols % col_div) ? num_cols + (col_div - num_cols % col_div) : num_cols;
}

/* io/data_formatter.h:36-51  adj_data[i] = 1.0 / colcount[col]  (double divide,
 * float store). */
void orc_normalize_by_outdegree(uint32_t num_rows, uint32_t num_cols,
                                const uint32_t *indptr, const uint32_t *indices, float *data)
{
    uint32_t nnz = indptr[num_rows];
    uint32_t *cnt = (uint32_t *)calloc(num_cols ? num_cols : 1, sizeof(uint32_t));
    for (uint32_t n = 0; n < nnz; n++) cnt[indices[n]]++;
    for (uint32_t r = 0; r < num_rows; r++)
        for (uint32_t i = indptr[r]; i < indptr[r + 1]; i++)
            data[i] = (float)(1.0 / cnt[indices[i]]);
    free(cnt);
}

/* app/sssp.h:16-62  _preprocess, restated WITH the reference's in-place
 * aliasing behaviour.  The reference sets every weight to 1 (:18-21), then
 * walks the rows inserting a weight-0 self edge into the live std::vectors
 * (:30-61).  Per row it reads
 *     start = adj_indptr[row]      -- already rewritten at :60, i.e. shifted by
 *                                     k = number of insertions made so far
 *     end   = adj_indptr[row + 1]  -- still the ORIGINAL, unshifted value
 * while the row's entries actually sit at [orig_start+k, orig_end+k).  So the
 * scan window covers only the first (len - k) entries of the row:
 *   window == 0 (start == end, :33-36): self edge inserted in front of the row
 *                                       (an existing diagonal keeps weight 1);
 *   window  < 0 (start  > end):         loop body never runs, row unchanged;
 *   window  > 0: first entry with col == row gets weight 0 (:41-43), else the
 *                self edge goes in front of the first entry with col > row
 *                (:44-48) or in front of the LAST WINDOW entry (:49-53).
 * Once k exceeds a row's length that row gets no self edge at all; faithful
 * parity with the reference requires reproducing exactly that.
 * Output arrays must hold nnz + num_rows entries; returns the new nnz.
 * tests/ pins this against a literal list-insert simulation of :16-62. */
uint32_t orc_sssp_preprocess(uint32_t num_rows,
                             const uint32_t *indptr, const uint32_t *indices,
                             uint32_t *out_indptr, uint32_t *out_indices, float *out_data)
{
    uint32_t w = 0;
    int64_t k = 0; /* insertions so far */
    out_indptr[0] = 0;
    for (uint32_t r = 0; r < num_rows; r++) {
        uint32_t start = indptr[r], end = indptr[r + 1];
        int64_t len = (int64_t)end - (int64_t)start;
        int64_t win = len - k;
        int64_t ins = -1, zero_at = -1;
        if (win == 0) {
            ins = 0;
        } else if (win > 0) {
            for (int64_t j = 0; j < win; j++) {
                uint32_t c = indices[start + j];
                if (c == r) { zero_at = j; break; }
                else if (c > r) { ins = j; break; }
                else if (j == win - 1) { ins = j; break; }
            }
        }
        for (int64_t j = 0; j < len; j++) {
            if (j == ins) { out_indices[w] = r; out_data[w] = 0.0f; w++; }
            out_indices[w] = indices[start + j];
            out_data[w] = (j == zero_at) ? 0.0f : 1.0f;
            w++;
        }
        if (ins == 0 && len == 0) { out_indices[w] = r; out_data[w] = 0.0f; w++; }
        if (ins >= 0) k++;
        out_indptr[r + 1] = w;
    }
    return w;
}

/* graphlily/global.h:153-164  convert_sparse_vec_to_dense_vec */
void orc_sparse_to_dense(const orc_idx_val_t *sv, uint32_t range, float zero, float *dense)
{
    for (uint32_t i = 0; i < range; i++) dense[i] = zero;
    int nnz = (int)sv[0].index;
    for (int i = 1; i < nnz + 1; i++) dense[sv[i].index] = sv[i].val;
}

/* ---------------------------------------------------------------- SpMV --- */

/* module/spmv_module.h:478-510  compute_reference_results(vector) */
void orc_spmv(int op, float zero, uint32_t num_rows,
              const uint32_t *indptr, const uint32_t *indices, const float *data,
              const float *x, float *y)
{
    for (uint32_t r = 0; r < num_rows; r++) y[r] = zero;
    switch (op) {
    case ORC_MULADD:
        for (uint32_t r = 0; r < num_rows; r++) {
            for (uint32_t i = indptr[r]; i < indptr[r + 1]; i++)
                y[r] += data[i] * x[indices[i]];
        }
        break;
    case ORC_ANDOR:
        for (uint32_t r = 0; r < num_rows; r++) {
            for (uint32_t i = indptr[r]; i < indptr[r + 1]; i++)
                y[r] = (float)(y[r] || (data[i] && x[indices[i]]));
        }
        break;
    case ORC_ADDMIN:
        for (uint32_t r = 0; r < num_rows; r++) {
            for (uint32_t i = indptr[r]; i < indptr[r + 1]; i++) {
                float t = data[i] + x[indices[i]];
                /* std::min(a, b) == (b < a) ? b : a */
                y[r] = (t < y[r]) ? t : y[r];
            }
        }
        break;
    default:
        break; /* reference prints "Invalid semiring" and returns the zero-filled vector */
    }
}

/* module/spmv_module.h:513-532  compute_reference_results(vector, mask):
 * WriteToZero -> rows with mask != 0 become literal 0; EVERY other mask type
 * (including kNoMask) -> rows with mask == 0 become literal 0. */
void orc_spmv_masked(int op, float zero, int mask_type, uint32_t num_rows,
                     const uint32_t *indptr, const uint32_t *indices, const float *data,
                     const float *x, const float *mask, float *y)
{
    orc_spmv(op, zero, num_rows, indptr, indices, data, x, y);
    if (mask_type == ORC_WRITETOZERO) {
        for (uint32_t i = 0; i < num_rows; i++) if (mask[i] != 0) y[i] = 0;
    } else {
        for (uint32_t i = 0; i < num_rows; i++) if (mask[i] == 0) y[i] = 0;
    }
}

/* -------------------------------------------------------------- SpMSpV --- */

/* module/spmspv_module.h:445-520  compute_reference_results(vector, mask):
 * dense result; (min,+) product saturates at FLOAT_INF; masked-off rows become
 * semiring.zero and the mask is compared against semiring.zero. */
void orc_spmspv(int op, float zero, int mask_type, uint32_t num_rows,
                const uint32_t *csc_indptr, const uint32_t *csc_indices, const float *csc_data,
                const orc_idx_val_t *v, const float *mask, float *y)
{
    uint32_t vnnz = v[0].index;
    for (uint32_t r = 0; r < num_rows; r++) y[r] = zero;
    for (uint32_t a = 0; a < vnnz; a++) {
        float xv = v[a + 1].val;
        uint32_t col = v[a + 1].index;
        for (uint32_t e = csc_indptr[col]; e < csc_indptr[col + 1]; e++) {
            uint32_t row = csc_indices[e];
            float m = csc_data[e];
            float incr;
            switch (op) {
            case ORC_MULADD:
                incr = m * xv;
                y[row] += incr;
                break;
            case ORC_ANDOR:
                incr = (float)(m && xv);
                y[row] = (float)(y[row] || incr);
                break;
            case ORC_ADDMIN:
                if (m > ORC_FLOAT_INF || xv > ORC_FLOAT_INF) {
                    incr = ORC_FLOAT_INF;
                } else {
                    incr = m + xv;
                    if (incr > ORC_FLOAT_INF) incr = ORC_FLOAT_INF;
                }
                y[row] = (y[row] < incr) ? y[row] : incr;
                break;
            default:
                break;
            }
        }
    }
    for (uint32_t i = 0; i < num_rows; i++) {
        int off;
        switch (mask_type) {
        case ORC_NOMASK:      off = 0; break;
        case ORC_WRITETOONE:  off = (mask[i] == zero); break;
        case ORC_WRITETOZERO: off = (mask[i] != zero); break;
        default:              off = 1; break;
        }
        if (off) y[i] = zero;
    }
}

/* --------------------------------------------------------------- apply --- */

/* module/add_scalar_vector_dense_module.h:195-204 */
void orc_ewise_add(const float *in, uint32_t len, float val, float *out)
{
    for (uint32_t i = 0; i < len; i++) out[i] = in[i] + val;
}

/* module/assign_vector_dense_module.h:223-246; returns -1 for kNoMask (the
 * reference prints "Invalid mask type" and exits). */
int orc_assign_dense(int mask_type, const float *mask, float *inout, uint32_t len, float val)
{
    if (mask_type == ORC_WRITETOZERO) {
        for (uint32_t i = 0; i < len; i++) if (mask[i] == 0) inout[i] = val;
    } else if (mask_type == ORC_WRITETOONE) {
        for (uint32_t i = 0; i < len; i++) if (mask[i] != 0) inout[i] = val;
    } else {
        return -1;
    }
    return 0;
}

/* module/assign_vector_sparse_module.h:306-315 (BFS mode) */
void orc_assign_sparse(const orc_idx_val_t *mask, float *inout, float val)
{
    for (uint32_t i = 0; i < mask[0].index; i++) inout[mask[i + 1].index] = val;
}

/* module/assign_vector_sparse_module.h:318-335 (SSSP mode): relax + emit the
 * new frontier in mask order with head {count, 0}. new_frontier must hold
 * mask[0].index + 1 entries. */
void orc_assign_sparse_new_frontier(const orc_idx_val_t *mask, float *inout,
                                    orc_idx_val_t *new_frontier)
{
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < mask[0].index; i++) {
        if (inout[mask[i + 1].index] > mask[i + 1].val) {
            inout[mask[i + 1].index] = mask[i + 1].val;
            new_frontier[1 + cnt++] = mask[i + 1];
        }
    }
    new_frontier[0].index = cnt;
    new_frontier[0].val = 0;
}

/* ---------------------------------------------------------------- apps --- */

/* app/bfs.h:350-360  BFS::compute_reference_results.  SpMV is Logical with
 * kMaskWriteToZero (bfs.h:55-56), DenseAssign is kMaskWriteToOne (bfs.h:60).
 * Result: level+1, 0 = unreached. */
void orc_bfs(uint32_t n, const uint32_t *indptr, const uint32_t *indices, const float *data,
             uint32_t source, uint32_t num_iterations, float *distance)
{
    float *input = (float *)malloc((size_t)n * sizeof(float));
    float *tmp = (float *)malloc((size_t)n * sizeof(float));
    for (uint32_t i = 0; i < n; i++) { input[i] = 0.0f; distance[i] = 0.0f; }
    input[source] = 1; distance[source] = 1;
    for (uint32_t iter = 1; iter <= num_iterations; iter++) {
        orc_spmv_masked(ORC_ANDOR, 0.0f, ORC_WRITETOZERO, n, indptr, indices, data, input, distance, tmp);
        memcpy(input, tmp, (size_t)n * sizeof(float));
        orc_assign_dense(ORC_WRITETOONE, input, distance, n, (float)(iter + 1));
    }
    free(input); free(tmp);
}

/* app/pagerank.h:150-159  PageRank::compute_reference_results.  rank init is
 * float(1.0 / n) (:151), teleport is (1 - damping) / n evaluated in float
 * (:156, damping is a float parameter). n is the PADDED row count. */
void orc_pagerank(uint32_t n, const uint32_t *indptr, const uint32_t *indices, const float *data,
                  float damping, uint32_t num_iterations, float *rank)
{
    float *tmp = (float *)malloc((size_t)n * sizeof(float));
    float init = (float)(1.0 / n);
    float teleport = (1 - damping) / n;
    for (uint32_t i = 0; i < n; i++) rank[i] = init;
    for (uint32_t iter = 1; iter <= num_iterations; iter++) {
        orc_spmv(ORC_MULADD, 0.0f, n, indptr, indices, data, rank, tmp);
        orc_ewise_add(tmp, n, teleport, rank);
    }
    free(tmp);
}

/* app/sssp.h:246-253  SSSP::compute_reference_results: repeated (min,+) SpMV,
 * no mask; `zero` is the semiring's "infinity" (global.h:99 uses 255). */
void orc_sssp(uint32_t n, const uint32_t *indptr, const uint32_t *indices, const float *data,
              float zero, uint32_t source, uint32_t num_iterations, float *dist)
{
    float *input = (float *)malloc((size_t)n * sizeof(float));
    for (uint32_t i = 0; i < n; i++) input[i] = zero;
    input[source] = 0;
    for (uint32_t iter = 1; iter <= num_iterations; iter++) {
        orc_spmv(ORC_ADDMIN, zero, n, indptr, indices, data, input, dist);
        memcpy(input, dist, (size_t)n * sizeof(float));
    }
    if (num_iterations == 0) memcpy(dist, input, (size_t)n * sizeof(float));
    free(input);
}

/* -------------------------------------------- CPU baseline (bench only) --- */

/* Row-parallel variant of orc_spmv for bench.py's "all host cores" baseline
 * line.  Same per-row arithmetic and order as orc_spmv; rows are
 * independent so results are identical.  Compiled with -fopenmp. */
void orc_spmv_omp(int op, float zero, uint32_t num_rows,
                  const uint32_t *indptr, const uint32_t *indices, const float *data,
                  const float *x, float *y)
{
    #pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t r = 0; r < (int64_t)num_rows; r++) {
        float acc = zero;
        for (uint32_t i = indptr[r]; i < indptr[r + 1]; i++) {
            float a = data[i], b = x[indices[i]];
            if (op == ORC_MULADD) acc += a * b;
            else if (op == ORC_ANDOR) acc = (float)(acc || (a && b));
            else { float t = a + b; acc = (t < acc) ? t : acc; }
        }
        y[r] = acc;
    }
}

/* ============================================================================================================
 * The reference's other value types (graphlily/global.h:62-64): `unsigned` and ap_ufixed<32, 8, AP_RND, AP_SAT>.
 *
 * PARITY UNPINNED for this section: the reference has no CPU path in these types -- its compute_reference_results
 * members always compute in float (spmv_module.h:478-532 works on the float copy of the matrix) and its tests compare
 * float(kernel result) with that within 1e-4 (test_module_spmv_spmspv.cpp:33-40).  What is restated here is the
 * arithmetic of the device ALUs, hw/ufixed_pe_fwd.h:23-65 (pe_ufixed_mul_alu: a * b / a && b / a + b;
 * pe_ufixed_add_alu: a + b / a || b / MIN(a, b)), applied in the loop order of the float reference, with the
 * assignment semantics of the value type written out:
 *   unsigned                 C arithmetic, results wrap mod 2^32; a && b and a || b are 0 / 1.
 *   ap_ufixed<32,8,RND,SAT>  word w stands for w / 2^24.  A product has 48 fraction bits and is assigned with
 *                            AP_RND (add half an ulp, truncate: round half up) and AP_SAT (clamp to 2^32 - 1); a sum
 *                            has the operands' fraction bits and is only clamped; a && b and a || b are 0 / 1.0 = 1 << 24.
 * Words travel as uint32_t; a sparse element is {uint32 index; uint32 value}.
 * ============================================================================================================ */
enum { ORC_VAL_FLOAT = 0, ORC_VAL_UNSIGNED = 1, ORC_VAL_UFIXED_32_8 = 2 };
typedef struct { uint32_t index; uint32_t val; } orc_idx_word_t;

static uint32_t orc_w_one(int vt) { return vt == ORC_VAL_UFIXED_32_8 ? (1u << 24) : 1u; }

/* pe_ufixed_mul_alu (hw/ufixed_pe_fwd.h:27-45) followed by the assignment to ValT */
static uint32_t orc_w_mul(int op, int vt, uint32_t a, uint32_t b)
{
    switch (op) {
    case ORC_MULADD:
        if (vt == ORC_VAL_UNSIGNED) return a * b;
        {
            uint64_t p = (uint64_t)a * (uint64_t)b;        /* 16 integer bits, 48 fraction bits */
            uint64_t r = (p + (1ull << 23)) >> 24;         /* AP_RND to 24 fraction bits */
            return r > 0xffffffffull ? 0xffffffffu : (uint32_t)r;   /* AP_SAT */
        }
    case ORC_ANDOR:
        return (a != 0 && b != 0) ? orc_w_one(vt) : 0u;
    default: /* ORC_ADDMIN */
        if (vt == ORC_VAL_UNSIGNED) return a + b;
        {
            uint64_t s = (uint64_t)a + (uint64_t)b;
            return s > 0xffffffffull ? 0xffffffffu : (uint32_t)s;
        }
    }
}

/* pe_ufixed_add_alu (hw/ufixed_pe_fwd.h:47-65) followed by the assignment to ValT */
static uint32_t orc_w_add(int op, int vt, uint32_t a, uint32_t b)
{
    switch (op) {
    case ORC_MULADD:
        if (vt == ORC_VAL_UNSIGNED) return a + b;
        {
            uint64_t s = (uint64_t)a + (uint64_t)b;
            return s > 0xffffffffull ? 0xffffffffu : (uint32_t)s;
        }
    case ORC_ANDOR:
        return (a != 0 || b != 0) ? orc_w_one(vt) : 0u;
    default:
        return a < b ? a : b;
    }
}

/* float -> value word, the conversion csr_matrix_convert_from_float<val_t> performs (io/data_loader.h:75-84):
 * unsigned: C conversion (truncation toward zero); ap_ufixed: AP_RND / AP_SAT (negative values clamp to 0) */
uint32_t orc_word_from_float(int vt, float v)
{
    if (vt == ORC_VAL_UNSIGNED) return v <= 0.0f ? 0u : (v >= 4294967296.0f ? 0xffffffffu : (uint32_t)v);
    if (!(v > 0.0f)) return 0u;
    double q = (double)v * 16777216.0 + 0.5;
    if (q >= 4294967296.0) return 0xffffffffu;
    return (uint32_t)q;   /* floor: q > 0 */
}

float orc_word_to_float(int vt, uint32_t w)
{
    return vt == ORC_VAL_UNSIGNED ? (float)w : (float)((double)w / 16777216.0);
}

/* SpMVModule::compute_reference_results (module/spmv_module.h:478-532) in the value type: accumulator initialised to
 * zero, sequential CSR order; masked-off rows are the literal 0 and the mask is compared with 0 */
void orc_spmv_words(int op, int vt, uint32_t zero, int mask_type, uint32_t num_rows,
                    const uint32_t *indptr, const uint32_t *indices, const uint32_t *data,
                    const uint32_t *x, const uint32_t *mask, uint32_t *y)
{
    for (uint32_t r = 0; r < num_rows; r++) {
        uint32_t acc = zero;
        for (uint32_t i = indptr[r]; i < indptr[r + 1]; i++)
            acc = orc_w_add(op, vt, acc, orc_w_mul(op, vt, data[i], x[indices[i]]));
        if (mask_type == ORC_WRITETOZERO) { if (mask[r] != 0) acc = 0; }
        else if (mask_type != ORC_NOMASK) { if (mask[r] == 0) acc = 0; }
        y[r] = acc;
    }
}

/* SpMSpVModule::compute_reference_results (module/spmspv_module.h:445-520) in the value type: dense result; the mask
 * is compared with `zero` and masked-off rows are `zero` */
void orc_spmspv_words(int op, int vt, uint32_t zero, int mask_type, uint32_t num_rows,
                      const uint32_t *csc_indptr, const uint32_t *csc_indices, const uint32_t *csc_data,
                      const orc_idx_word_t *v, const uint32_t *mask, uint32_t *y)
{
    for (uint32_t r = 0; r < num_rows; r++) y[r] = zero;
    uint32_t active = v[0].index;
    for (uint32_t k = 1; k <= active; k++) {
        uint32_t col = v[k].index, xv = v[k].val;
        for (uint32_t e = csc_indptr[col]; e < csc_indptr[col + 1]; e++) {
            uint32_t row = csc_indices[e];
            y[row] = orc_w_add(op, vt, y[row], orc_w_mul(op, vt, csc_data[e], xv));
        }
    }
    for (uint32_t r = 0; r < num_rows; r++) {
        int off = 0;
        if (mask_type == ORC_WRITETOONE) off = (mask[r] == zero);
        else if (mask_type == ORC_WRITETOZERO) off = (mask[r] != zero);
        if (off) y[r] = zero;
    }
}

/* kernel_add_scalar_vector_dense (hw/kernel_add_scalar_vector_dense_impl.h:6-27): out = in + val in the value type */
void orc_ewise_add_words(int vt, const uint32_t *in, uint32_t len, uint32_t val, uint32_t *out)
{
    for (uint32_t i = 0; i < len; i++) out[i] = orc_w_add(ORC_MULADD, vt, in[i], val);
}

/* kernel_assign_vector_dense (hw/kernel_assign_vector_dense_impl.h:8-47) */
void orc_assign_dense_words(int mask_type, const uint32_t *mask, uint32_t *inout, uint32_t len, uint32_t val)
{
    for (uint32_t i = 0; i < len; i++)
        if ((mask_type == ORC_WRITETOZERO) ? (mask[i] == 0) : (mask[i] != 0)) inout[i] = val;
}

/* kernel_assign_vector_sparse_new_frontier (hw/kernel_assign_vector_sparse_new_frontier_impl.h:4-78): both value
 * types order like their words */
void orc_assign_sparse_new_frontier_words(const orc_idx_word_t *mask, uint32_t *inout, orc_idx_word_t *new_frontier)
{
    uint32_t n = mask[0].index, cnt = 0;
    for (uint32_t k = 1; k <= n; k++) {
        if (inout[mask[k].index] > mask[k].val) {
            inout[mask[k].index] = mask[k].val;
            new_frontier[++cnt] = mask[k];
        }
    }
    new_frontier[0].index = cnt;
    new_frontier[0].val = 0;
}
