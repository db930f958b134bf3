/*
 * graphlily_hip.h -- C ABI of libgraphlily_hip.so, the MI355X (gfx950) backend
 * behind the graphlily::module operator API.
 *
 * This is the drop-in boundary: everything the reference does by talking to
 * the FPGA through OpenCL/XRT -- cl::Buffer + enqueueMigrateMemObjects,
 * enqueueCopyBuffer, and enqueueTask(overlay, mode=1..6) -- is replaced by the
 * entry points below.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *  - Every function returns GL_OK (0) or a negative gl_status; the message of
 *    the last failure on the calling thread is available via gl_last_error().
 *    Nothing throws across the boundary.  (The reference prints and exits,
 *    xrt/includes/xcl2/xcl2.hpp:40-46; the C++ module layer in
 *    include/graphlily/ keeps that convention on top of these codes.)
 *  - `d_` parameters are DEVICE pointers (from gl_buf_alloc or any other HIP
 *    allocation on the current device, e.g. a torch tensor's data_ptr()).
 *  - Work is enqueued on the library's current stream (gl_set_stream) and is
 *    asynchronous; gl_sync() is the analogue of command_queue_.finish().
 *  - val_t is float (graphlily/global.h:64), idx_t is uint32_t (:65), a
 *    sparse-vector element is {uint32 index; float val} with element [0]
 *    holding the non-zero count in .index (global.h:69, spmspv_module.h:53-60).
 *  - There is no CPU fallback: without a HIP device gl_init fails and every
 *    compute entry point returns GL_ERR_NOT_INITIALIZED.
 */
#ifndef GRAPHLILY_HIP_H_
#define GRAPHLILY_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gl_status {
    GL_OK = 0,
    GL_ERR_INVALID_ARG = -1,
    GL_ERR_HIP = -2,
    GL_ERR_NOT_INITIALIZED = -3,
    GL_ERR_IO = -4,
    GL_ERR_UNSUPPORTED = -5
} gl_status;

/* graphlily/global.h:83-87 OperationType */
typedef enum gl_op { GL_OP_MULADD = 0, GL_OP_ANDOR = 1, GL_OP_ADDMIN = 2 } gl_op;
/* graphlily/global.h:103-107 MaskType */
typedef enum gl_mask { GL_NOMASK = 0, GL_MASK_WRITETOZERO = 1, GL_MASK_WRITETOONE = 2 } gl_mask;

/* graphlily/global.h:69 idx_val_t (val_t = float) */
typedef struct gl_idx_val { uint32_t index; float val; } gl_idx_val;

typedef struct gl_spmv_plan_s *gl_spmv_plan;     /* formatted matrix for SpMV   */
typedef struct gl_spmspv_plan_s *gl_spmspv_plan; /* formatted matrix for SpMSpV */

/* ------------------------------------------------------------------ runtime
 * Replaces BaseModule::set_up_runtime / ModuleCollection::set_up_runtime
 * (module/base_module.h:106-133, app/module_collection.h:69-114): device
 * discovery, context, command queue.  No bitstream to load. */
int gl_init(int device);                /* hipSetDevice + library stream (+ gl_host_bind_near_device unless GRAPHLILY_BIND_NUMA=0) */
/* Restrict the CALLING thread -- and the threads it creates from now on, the library's own host team included -- to the CPUs of
 * the device's NUMA node (sysfs: the PCI device's numa_node, the node's cpulist), within the affinity mask the process had at
 * the library's first call.  The host half of a blocking call (the wait, the expansion of a packed result, copies out of
 * page-locked memory) costs 40-100 us more from the far socket of a two-socket host.  *numa_node / *cpus (may be NULL): what was
 * applied, -1 / 0 when nothing was (one node, sysfs silent, or no CPU of the node allowed).  Never an error. */
int gl_host_bind_near_device(int *numa_node, int *cpus);
int gl_device_count(int *count);
int gl_set_stream(void *hip_stream);    /* adopt a caller-owned hipStream_t; NULL is HIP's default (null) stream */
int gl_reset_stream(void);              /* go back to the library-owned stream  */
int gl_sync(void);                      /* command_queue_.finish()              */
/* A launch sequence of this library recorded once and replayed with one call (hipGraph): between
 * gl_graph_begin_capture and gl_graph_end_capture the calls on the library's stream are recorded instead of executed
 * (only asynchronous entry points may be used: no uploads / downloads / gl_sync / plan creation, and every buffer
 * the sequence needs must exist already -- run it once eagerly first).  The iterative drivers are launch-bound on
 * small frontiers (a BFS push iteration is ~10 launches of a few microseconds each). */
typedef struct gl_graph_s *gl_graph;
int gl_graph_begin_capture(void);
int gl_graph_end_capture(gl_graph *graph);
int gl_graph_launch(gl_graph graph);    /* async, on the library's stream */
int gl_graph_destroy(gl_graph graph);
const char *gl_last_error(void);
const char *gl_version(void);

/* ------------------------------------------------------------------ buffers
 * Replace cl::Buffer(CL_MEM_USE_HOST_PTR) + enqueueMigrateMemObjects
 * (e.g. module/spmv_module.h:424-439, :229-253) and
 * BaseModule::copy_buffer_device_to_device (module/base_module.h:82-85). */
int gl_buf_alloc(void **d_ptr, size_t bytes);
int gl_buf_free(void *d_ptr);
int gl_buf_h2d(void *d_dst, const void *h_src, size_t bytes);   /* blocking */
int gl_buf_d2h(void *h_dst, const void *d_src, size_t bytes);   /* blocking */
/* the same copy enqueued on the library stream without waiting (h_dst page-locked, gl_host_alloc: it then runs behind the
 * kernels already enqueued, with no host round trip in between; gl_sync before h_dst is read) */
int gl_buf_d2h_async(void *h_dst, const void *d_src, size_t bytes);
/* A BFS result (levels: floats holding small integers) read back PACKED: gl_levels_pack writes n levels (n a multiple of 8,
 * 16-byte aligned buffers) as bytes (bits = 8: levels 0 ... 255) or nibbles (bits = 4: 0 ... 15) into d_out, followed -- on the
 * next 16-byte boundary -- by tail_words raw words of d_tail (the schedule's control words: one read-back fetches both);
 * the caller copies n * bits / 8 bytes instead of 4 n over PCIe and gl_sync_levels_unpack turns them into the floats the
 * reference's send_*_device_to_host returns, on a few host threads: gl_sync + the expansion with the thread team started
 * BEFORE the wait (its master waits for the library's stream, the others spin until it returns -- a team woken ahead of a
 * 0.3 ms wait is asleep again when the wait ends).  (12 MB of levels: 225 us of PCIe become 28 - 55 us + ~30 us of host
 * work; larger values do not survive -- the drivers use this only when the iteration count allows, and only while it
 * measures faster than the float copy on the box at hand.) */
int gl_levels_pack(const float *d_levels, uint32_t n, int bits, const uint32_t *d_tail, uint32_t tail_words, void *d_out);
int gl_host_unpack_threads(void);   /* how many threads gl_sync_levels_unpack uses for a large vector (packing pays from 4 on) */
int gl_sync_levels_unpack(float *h_dst, const void *h_src, size_t n, int bits);
int gl_host_levels_unpack(float *h_dst, const void *h_src, size_t n, int bits);   /* the expansion alone (needs no GPU) */
/* The same read-back STREAMED (round 6): gl_levels_pack_stream's kernel stores the packed words straight into a page-locked host
 * block (gl_host_alloc of gl_levels_stream_bytes) in chunks of GL_LEVELS_CHUNK_WORDS, raising a flag word behind each chunk
 * (system-scope release) and behind the tail words; gl_sync_levels_unpack_stream starts the team, whose master waits for the
 * library's stream while the others expand chunk after chunk as the flags come up -- the expansion overlaps the PCIe transfer
 * instead of following it -- and copies the tail words out.  gl_levels_stream_arm clears the flags: call it on the host BEFORE
 * the pack (or the graph that holds it) is launched, once per run; the block belongs to one run at a time.  Block layout, in
 * 32-bit words: packed levels | tail at the next multiple of 4 | flags at the next multiple of 16, one cache line each. */
#define GL_LEVELS_CHUNK_WORDS 2048u
#define GL_LEVELS_FLAG_STRIDE_WORDS 16u
int gl_levels_stream_bytes(uint32_t n, int bits, uint32_t tail_words, size_t *bytes);
int gl_levels_stream_arm(void *h_block, uint32_t n, int bits, uint32_t tail_words);
int gl_levels_pack_stream(const float *d_levels, uint32_t n, int bits, const uint32_t *d_tail, uint32_t tail_words, void *h_block);
int gl_sync_levels_unpack_stream(float *h_dst, const void *h_block, size_t n, int bits, uint32_t *h_tail, uint32_t tail_words);
/* Blocking read-back of n floats that are EXPECTED to be BFS levels no larger than max_level (a caller that has seen only
 * level-writing kernels touch the buffer): packed on the device with every value checked, copied as nibbles / bytes, expanded
 * on host threads; if any value is not a small non-negative integer the floats themselves are copied -- the result is always
 * exactly the buffer's contents.  *packed (may be NULL) tells which way it went.  GRAPHLILY_BFS_U8=0: always the floats. */
int gl_buf_d2h_levels(float *h_dst, const float *d_src, size_t n, float max_level, int *packed);
int gl_buf_d2d(void *d_dst, const void *d_src, size_t bytes);   /* async    */
int gl_buf_fill_f32(float *d_dst, float value, size_t count);   /* async    */
int gl_buf_fill_u32(uint32_t *d_dst, uint32_t value, size_t count);   /* async    */
/* page-locked host memory for result read-back at full PCIe rate (the reference's host mirrors are
 * 4 KiB-aligned for the same reason, xcl2.hpp:61-76) */
int gl_host_alloc(void **h_ptr, size_t bytes);
int gl_host_free(void *h_ptr);
/* Block recycling.  The reference's drivers make a fresh cl::Buffer and a fresh 4 KiB-aligned host vector for
 * every send_*_host_to_device / send_*_device_to_host (app/bfs.h:107-113, xcl2.hpp:61-76); hipMalloc, hipFree and
 * hipHostMalloc of 12 MB cost more than the SpMV they serve.  gl_buf_alloc / gl_buf_free therefore recycle device
 * blocks by size (reuse is ordered by the library's stream), and gl_host_pool_alloc / gl_host_pool_free do the
 * same for 4 KiB-aligned host blocks of 64 KiB and more (no mmap + page faults per vector; they work before
 * gl_init: the C++ layer's aligned_allocator sits on them).  Device blocks up to 64 MB are carved from 256 MB slabs; a
 * request is served by the smallest parked block of its size .. +25 %; a slab whose blocks have all come back is carved
 * from its start again (and released if another one exists); moving to another device (gl_init) releases what is parked
 * and retires the slabs still in use.  Host blocks are plain pages -- on the MI355X box
 * pageable and page-locked copies run at the same 56 GB/s while page-locking 12 MB costs 2.5 ms
 * (profiles/r02_ubench_host.txt).  gl_pool_trim returns every cached block. */
/* gl_host_pool_reserve: make sure `count` paged-in blocks for requests of `bytes` are parked (the C++ modules do this when a
 * matrix is sent: a driver call then finds its n-element vectors -- two inputs, the result, the previous call's result still
 * alive -- without a miss, which costs 2-4 ms of page faults for 12 MB inside a 2 ms BFS).  Blocks of 2 MB and more are
 * 2 MB-aligned and advised as huge pages.
 * gl_host_fill_u32 / gl_host_sparse_to_dense: the two host loops the reference's drivers run on n-element vectors between
 * module calls -- vector fill and convert_sparse_vec_to_dense_vec (graphlily/global.h:153-164; the push -> pull switch of
 * app/bfs.h:196-201) -- on a few host threads (the header-only C++ layer calls them for large vectors; the library is
 * built with OpenMP, the caller need not be).  Need no GPU. */
int gl_host_pool_reserve(size_t bytes, uint32_t count);
int gl_host_fill_u32(void *h_dst, uint32_t word, size_t count);
int gl_host_sparse_to_dense(const gl_idx_val *h_sparse, uint32_t range, uint32_t zero_bits, void *h_dense);
int gl_host_pool_alloc(void **h_ptr, size_t bytes);
int gl_host_pool_free(void *h_ptr);
int gl_pool_trim(void);
/* device pool counters for tests and leak hunting: blocks currently handed out by gl_buf_alloc, bytes parked for reuse,
 * number of 256 MB slabs the small blocks are carved from (any pointer may be NULL) */
int gl_pool_stats(uint64_t *live_blocks, uint64_t *cached_bytes, uint32_t *slabs);

/* --------------------------------------------------------------------- SpMV
 * gl_spmv_plan_create replaces SpMVModule::load_and_format_matrix +
 * send_matrix_host_to_device (module/spmv_module.h:281-420): it re-lays the
 * host CSR (io/data_loader.h:18-30) out as the CDNA4 row-block / column-sorted
 * stream (gl_spmv.hip) and uploads it.  [row_begin,row_end) selects the row shard this device owns
 * (0,num_rows = whole matrix); x is always indexed by global column and y by
 * global row, so a shard writes y[row_begin..row_end) of a full-length y. */
int gl_spmv_plan_create(gl_spmv_plan *plan,
                        uint32_t num_rows, uint32_t num_cols,
                        const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data,
                        uint32_t row_begin, uint32_t row_end);
/* Same with planning hints.  GL_PLAN_NO_MULADD: the plan will only be run with the (||,&&) and (min,+)
 * semirings, whose LDS accumulators are 4 bytes -- the freed LDS holds a larger hot-column table.
 * Running such a plan with (+,x) returns GL_ERR_UNSUPPORTED. */
#define GL_PLAN_NO_MULADD 1u
/* GL_PLAN_BOOLEAN: the plan will only be run with the (||,&&) semiring.  Only the sparsity pattern of the
 * non-zero-valued entries is kept (4 bytes per entry; `a && b` is false for a == 0), x is packed to one
 * bit per column once per run and kept in LDS, the row accumulators are bits.  Results are identical to the
 * general plan's.  Running such a plan with another semiring returns GL_ERR_UNSUPPORTED.  Implies
 * GL_PLAN_NO_MULADD when the general layout has to be used (more than 8 x 1 179 648 columns). */
#define GL_PLAN_BOOLEAN 2u
/* Matrices whose stored values are equal within every column (unweighted graphs, out-degree-normalised
 * PageRank matrices, bench_spmv's constant 1/num_rows) are detected at plan creation and kept as a
 * "pattern plan": 4 bytes per entry, the column value is folded into z[c] = colval[c] (x) x[c] by one
 * extra pass over x per run -- the same float products as the general layout, hence the same results.
 * GL_PLAN_KEEP_VALUES switches the detection off (8-byte {index,value} entries whatever the values are). */
#define GL_PLAN_KEEP_VALUES 4u
/* Where the plan is formatted.  By default matrices of a million non-zeros and more are formatted ON THE GPU
 * (gl_format.hip: the shard's CSR is uploaded once; column degrees, the column-constant test, the per-block column
 * sort -- one stable radix sort -- group packing and emission are kernels; the O(rows + columns) decisions stay on
 * the host), smaller ones on the host with OpenMP.  The two formatters produce byte-identical device arrays
 * (gl_spmv_plan_export, tests/test_gpu_format.py).  These two flags force one or the other; the environment
 * variable GRAPHLILY_PLAN_DEVICE=0/1 does the same for plans created without them. */
#define GL_PLAN_HOST_FORMAT 8u
#define GL_PLAN_DEVICE_FORMAT 16u
/* GL_PLAN_REFERENCE_ORDER: a DIAGNOSTIC layout.  The shard's CSR is kept as it is and gl_spmv_run evaluates
 * SpMVModule::compute_reference_results (module/spmv_module.h:478-532) the way the reference writes it: a thread per
 * row, the entries in CSR order, a float accumulator starting at `zero`, separately rounded float multiply and add (no
 * FMA).  Results are bit-equal to the reference's CPU loop by construction, for all three semirings and masks -- the
 * tests run it next to the fast layouts to show that summation order (and the f64 accumulator of (+,x)) is the only
 * thing in which they differ from the reference.  Slow on hub rows (one thread walks the whole row); float only. */
#define GL_PLAN_REFERENCE_ORDER 32u
int gl_spmv_plan_create_ex(gl_spmv_plan *plan,
                           uint32_t num_rows, uint32_t num_cols,
                           const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data,
                           uint32_t row_begin, uint32_t row_end, uint32_t flags);
int gl_spmv_plan_destroy(gl_spmv_plan plan);
/* What the planner made of the matrix: nnz held by this plan (its shard), device bytes of the formatted matrix, work units;
 * the decomposition (row blocks x column segments, tallest block, 64-entry groups); the device layout (8-byte {index,value}
 * entries, 4-byte pattern entries with per-column values, the (||,&&)-only bit layout, or the diagnostic reference-order CSR);
 * the hot-column cache (columns whose x value is kept in LDS, the non-zeros they serve, the cold/hot interleave: 0 = no hot
 * table, 5 = 3 cold + 3 hot groups per wavefront iteration); and how the plan refills its hot table / packed gather vector
 * per run: a gathering helper kernel, one streaming pass over x, or no helper launch at all (short streams
 * with a small hot table: the workgroups gather it themselves); packed_columns = length of the packed vector (0: cold
 * entries index x itself). */
#define GL_LAYOUT_GENERAL 0
#define GL_LAYOUT_PATTERN 1
#define GL_LAYOUT_BOOLEAN 2
#define GL_LAYOUT_REFERENCE_ORDER 3
#define GL_HELPER_GATHER 0
#define GL_HELPER_SPREAD 1
#define GL_HELPER_SELF_HOT 2
#define GL_HELPER_NONE 3
typedef struct gl_spmv_plan_desc {
    uint64_t nnz, device_bytes, groups, hot_nnz;
    uint32_t num_units, blocks, segments, max_block_rows, hot_columns, packed_columns;
    int layout, mix, helper;
} gl_spmv_plan_desc;
int gl_spmv_plan_describe(gl_spmv_plan plan, gl_spmv_plan_desc *out);
/* Debugging / tests: copy one of the plan's device arrays to the host.  `array` is one of GL_PLAN_ARRAY_*; *bytes
 * receives its size (also when h_dst is NULL or capacity is too small, in which case nothing is copied and
 * GL_ERR_INVALID_ARG is returned for a non-NULL h_dst). */
#define GL_PLAN_ARRAY_ENTRIES 0
#define GL_PLAN_ARRAY_BASES 1
#define GL_PLAN_ARRAY_UNITS 2
#define GL_PLAN_ARRAY_HUB_ROWS 3
#define GL_PLAN_ARRAY_SPANS 4
#define GL_PLAN_ARRAY_HOT 5     /* general / pattern layouts: the run-coded hot stream, */
#define GL_PLAN_ARRAY_HOT_HDR 6 /* its per-group run masks and bases, */
#define GL_PLAN_ARRAY_PRESENT 7 /* and every unit's list of the hot-table slots that occur in it */
int gl_spmv_plan_export(gl_spmv_plan plan, int array, void *h_dst, size_t capacity, size_t *bytes);
/* Extension for row-sharded (||,&&) runs: x as a bit vector supplied by the caller, so that ranks exchange
 * n/8 bytes per iteration instead of 4n (multi-GPU).  gl_spmv_plan_bits_words: length in 32-bit words
 * of the bit vector the boolean layout reads (whole 144 KB phases, 0 for the other layouts; bits past num_cols
 * must be 0).  gl_pack_bits: bits[i / 32] bit (i % 32) = (x[i] != 0) for i < n, whole 64-bit words are written
 * (d_bits 8-byte aligned).  gl_spmv_run_bits: gl_spmv_run of a GL_PLAN_BOOLEAN plan with op (||,&&) on that
 * bit vector (16-byte aligned) instead of a float x. */
int gl_spmv_plan_bits_words(gl_spmv_plan plan, uint64_t *words);
int gl_pack_bits(const float *d_x, uint32_t n, uint32_t *d_bits);
/* the inverse: x[i] = bit i ? 1.0f : 0.0f for i < n (a frontier kept as bits handed back as the float vector of the module API) */
int gl_unpack_bits(const uint32_t *d_bits, uint32_t n, float *d_x);
int gl_spmv_run_bits(gl_spmv_plan plan, const uint32_t *d_bits, const float *d_mask, float *d_y, float zero,
                     int mask_type);
/* Extension: one BFS pull iteration fused into one launch.  Equivalent to SpMVModule::run with the (||,&&) semiring
 * masked WriteToZero by `distance`, eWiseAddModule::run(+0) into the frontier vector and
 * AssignVectorDenseModule::run(level) WriteToOne by that vector (app/bfs.h:118-123), with the frontier kept as
 * bits: rows with an edge from the frontier `bits_in` whose distance is still 0 get distance = level and their bit
 * in `bits_out` (every word of the plan's row range is written).  Needs an unsplit GL_PLAN_BOOLEAN plan whose
 * row range starts on a multiple of 64 (GL_ERR_UNSUPPORTED otherwise: use the three calls).  Both bit vectors
 * have gl_spmv_plan_bits_words words, are 16-byte aligned and distinct. */
int gl_bfs_pull_step(gl_spmv_plan plan, const uint32_t *d_bits_in, uint32_t *d_bits_out, float *d_distance, float level);
/* gl_spmv_run replaces enqueueTask(overlay, mode = 1) (module/spmv_module.h:471-475,
 * hw/overlay.cpp:308-330 -> hw/kernel_spmv_impl.h:392-819):
 *   y[r] = mask_r ? ( zero (+) sum_{i in row r} A_i (x) x[col_i] ) : 0
 * op selects ((+),(x)): MULADD (+,*), ANDOR (||,&&), ADDMIN (min,+); `zero` is
 * SemiringType::zero (global.h:90-94).  d_mask may be NULL iff mask_type is
 * GL_NOMASK.  WriteToZero keeps rows with mask==0, WriteToOne rows with
 * mask!=0; masked-off rows are written as literal 0 (kernel_spmv_impl.h:361-386,
 * spmv_module.h:518-530). */
int gl_spmv_run(gl_spmv_plan plan, const float *d_x, const float *d_mask, float *d_y,
                int op, float zero, int mask_type);
/* Extension for iterative callers that feed every result straight back as the next vector -- PageRank::pull and SSSP::pull swap
 * their two buffers instead of copying (app/pagerank.h:84-88, app/sssp.h:157-165): while a plan is CHAINED, an unmasked float run's
 * epilogue also stores y in the form the next run reads x in (its slots of the plan's packed vector and hot table, times the
 * column's value in pattern plans), and a run whose d_x IS the previous run's d_y skips the helper launch that would build
 * them (8 us of orkut's 0.21 ms PageRank iteration).  The caller promises that such a vector has not been written in between;
 * results are bit-identical either way.  *active (may be NULL): whether the plan can chain at all -- the whole square matrix,
 * unsplit blocks, the streaming helper; otherwise the call changes nothing.  on = 0 ends the chain (always do so before the
 * vectors are handed to anything else). */
int gl_spmv_plan_chain(gl_spmv_plan plan, int on, int *active);

/* Measurement hook (bench.py roofline): between gl_prof_begin and gl_prof_end every launch of the
 * dominant SpMV kernel is bracketed by HIP events recorded on the stream it is launched on.
 * gl_prof_end synchronises and returns the summed kernel time and the number of launches. */
/* `every`: bracket only every n-th launch (0 or 1 = all).  An event pair keeps the neighbouring launches from overlapping
 * the kernel's first and last workgroups, so bracketing every launch slows a back-to-back sequence by ~5 %; sampling
 * keeps the timed region close to what it is without the profiler. */
int gl_prof_begin(uint32_t max_launches, uint32_t every);
int gl_prof_end(double *total_ms, uint32_t *launches);
/* GPU time of everything enqueued on the library's stream between the two calls (one HIP event pair; gl_span_end waits
 * for the second event): how long a whole launch sequence -- e.g. a replayed BFS schedule -- occupies the device,
 * without the read-back that follows it. */
int gl_span_begin(void);
int gl_span_end(double *ms);

/* ------------------------------------------------------------------- SpMSpV
 * gl_spmspv_plan_create replaces SpMSpVModule::load_and_format_matrix +
 * send_matrix_host_to_device (module/spmspv_module.h:263-370, formatCSC
 * io/data_formatter.h:543-721) for a host CSC (io/data_loader.h:92-104).
 * [row_begin,row_end) keeps only the entries whose row falls in the shard. */
int gl_spmspv_plan_create(gl_spmspv_plan *plan,
                          uint32_t num_rows, uint32_t num_cols,
                          const uint32_t *h_indptr, const uint32_t *h_indices, const float *h_data,
                          uint32_t row_begin, uint32_t row_end);
int gl_spmspv_plan_destroy(gl_spmspv_plan plan);
int gl_spmspv_plan_info(gl_spmspv_plan plan, uint64_t *nnz, uint64_t *device_bytes);

/* gl_spmspv_run replaces enqueueTask(overlay, mode = 2) (module/spmspv_module.h:436-441,
 * hw/kernel_spmspv_impl.h:448-562).  d_vector is a sparse vector (count in
 * [0].index, at most num_cols entries); d_result receives the sparse result:
 * [0] = {nnz, zero}, then every row r (ascending) with acc[r] != zero whose mask
 * allows it: NOMASK all, WriteToOne mask[r] != zero, WriteToZero mask[r] == zero
 * (hw/kernel_spmspv_impl.h:262-283 -- note: compared with `zero`, not 0).
 * d_result must hold rows_in_shard + 1 elements.  The (min,+) product saturates
 * at FLOAT_INF = 999999999 (hw/float_pe.h:24-33). */
int gl_spmspv_run(gl_spmspv_plan plan, const gl_idx_val *d_vector, const float *d_mask,
                  gl_idx_val *d_result, int op, float zero, int mask_type);
/* Blocks until the LAST gl_spmspv_run* enqueued on this plan has written its result list, and returns the result count
 * (SpMSpVModule::run is blocking, and get_results_nnz follows it in every push loop: module/spmspv_module.h:436-441,
 * :239-242).  The operator's last workgroup stores {sequence, count} to page-locked host memory once every result is in
 * device memory, so the host neither waits for the stream's completion signal nor copies the head element back (~5 us
 * + a 4-byte copy per call).  A run recorded into a graph keeps no record: the call is then gl_sync and *nnz = 0xffffffff
 * (read the head element with gl_sparse_nnz).  Work enqueued on OTHER streams
 * is not waited for. */
int gl_spmspv_wait(gl_spmspv_plan plan, uint32_t *nnz);
/* Extension: how many runs on this plan had a workgroup rendezvous time out so far (their result lists were emptied; a blocking
 * caller got GL_ERR_HIP from gl_spmspv_wait).  For callers whose runs keep no completion record -- runs recorded into a hipGraph
 * and replayed -- this is where such a failure surfaces: read it after the replay (waits for the stream, one 4-byte copy).  The
 * reference has no counterpart (its kernel cannot time out: module/spmspv_module.h:436-441 blocks in finish()). */
int gl_spmspv_failed_runs(gl_spmspv_plan plan, uint32_t *count);
/* Extension: gl_spmspv_run followed by gl_assign_sparse(d_result, d_inout, val) -- the push iteration of BFS
 * (app/bfs.h:146-148: SpMSpV, then AssignVectorSparse::run(val) with the result as its mask) -- with the assign
 * done by the pass that writes the result list (one launch and one read of the list less).  d_inout may be the
 * vector d_mask points to (BFS masks with the distances it assigns): an entry's mask word is read before its
 * own row is written and no other entry touches it.  d_inout == NULL is gl_spmspv_run. */
int gl_spmspv_run_assign(gl_spmspv_plan plan, const gl_idx_val *d_vector, const float *d_mask,
                         gl_idx_val *d_result, int op, float zero, int mask_type, float *d_inout, float val);

/* Extensions that take the host out of the BFS loop (the reference reads the result count back every push
 * iteration to decide the direction, app/bfs.h:180-190, and converts the frontier on the host at the switch, :195-205).
 * The frontier lives as BITS only; a driver enqueues the WHOLE run -- one launch per iteration slot -- and device-side
 * control words decide what every slot does.  No call synchronises or copies to the host; the sequence can be recorded once
 * (gl_graph_*) and replayed for any source.
 *   d_ctl     ctl_words >= 18 + 2 * slots words, 8-byte aligned: [0] first pull slot (0xffffffff while pushing), [1] push
 *             iterations of the first push phase (the reference's count), [2] source vertex (written before the run, e.g.
 *             gl_buf_fill_u32), [3] pushes after a pull step handed the loop back, [4] the slot that handed back, [5..14]
 *             internal, [15] ctl_words; behind them two arrays of S = (ctl_words - 16) / 2 words: [16 + s] the number of
 *             vertices slot s reached, [16 + S + s] how slot s was evaluated (1 scattered, 2 streamed row-wise, 3 bottom-up;
 *             slots >= S are not recorded).
 *   d_bits    nvec >= slots + 2 bit vectors of bits_words words each, contiguous, 16-byte aligned (bits_words a multiple of
 *             4, at least gl_spmv_plan_bits_words of the row plan): slot s (1, 2, ...) reads vector s and writes vector
 *             s + 1, which therefore holds exactly the vertices at distance s + 1 when the run is over.
 *   gl_bfs_bits_begin       distance[i] = (i == source), vector 1 = {source}, the others, the tallies behind them and the
 *             control words cleared; ctl[0] = first_pull_slot: 0xffffffff for pull_push (push until the rule says
 *             otherwise), 0 for a BFS that pulls in every slot (app/bfs.h:106-126).
 *   gl_bfs_bits_shard_step  slot `slot` (1-based) on the rows of `csc` / `rows` (the SpMSpV plan and the GL_PLAN_BOOLEAN SpMV
 *             plan of the same matrix and row range -- the whole matrix on one GPU, a rank's shard cut on multiples of 64
 *             rows otherwise; csrc/gl_bfs_shard.h).  The launch starts with the decision of slot - 1: every workgroup adds up
 *             all ranks' TALLIES of that slot -- vertices reached, their global column lengths (d_col_len: n words, the
 *             whole matrix'), their row lengths -- and replays the reference's loop condition (do { push } while (it < N &&
 *             new frontier / n < threshold), app/bfs.h:180-190, same float comparison) on a private copy of the control
 *             words; after a pull the opposite decision (back_threshold > 0 and a slot follows: push again -- an
 *             extension, distances do not depend on the direction).  Then the step runs as that state says: the scattering
 *             push straight into the next frontier's bits (SpMSpV (||,&&) masked by d_distance + AssignVectorSparse(level),
 *             no accumulator, no compaction), the streaming pull (gl_bfs_pull_step; also a push whose frontier's columns
 *             hold more than 1/128 of the non-zeros), or -- once the rows not reached yet hold less than a third of the
 *             non-zeros -- the bottom-up scan (a thread per unreached row looks for a neighbour in the frontier).  Every
 *             rank tallies what its step adds into its 256 bytes of d_tally; row-sharded drivers exchange them with the
 *             slot's bit vector (gl_dist_all_gather_bits_tally: one grouped operation) -- no reduction collective, no host.
 *             may_continue_prev: bit 0 "the reference's loop may go on after slot - 1", bit 1 "a slot follows".
 *             d_tally: GL_BFS_TALLY_WORDS(slots, world) words (behind the bit vectors: gl_bfs_bits_begin clears them);
 *             d_tally_in: where the previous slot's tallies of ALL ranks are read (NULL = d_tally).
 *   gl_bfs_bits_shard_finish  after the last slot: its decision, and the final control words into d_ctl (where the host
 *             reads the reference's push count and the per-slot records).
 * Each rank's d_distance is full-length but only its own rows are written: read back the slice. */
int gl_bfs_bits_begin(uint32_t *d_ctl, uint32_t ctl_words, float *d_distance, uint32_t n, uint32_t *d_bits, uint32_t bits_words,
                      uint32_t nvec, uint32_t first_pull_slot);
#define GL_BFS_TALLY_HEAD_WORDS 64
#define GL_BFS_TALLY_RANK_WORDS 64
#define GL_BFS_TALLY_WORDS(slots, world) (GL_BFS_TALLY_HEAD_WORDS + (size_t)(slots) * (size_t)(world) * GL_BFS_TALLY_RANK_WORDS)
int gl_bfs_bits_shard_step(gl_spmspv_plan csc, gl_spmv_plan rows, const uint32_t *d_bits_in, uint32_t *d_bits_out, uint32_t bits_words,
                           float *d_distance, float level, uint32_t *d_ctl, uint32_t *d_tally, const uint32_t *d_tally_in, uint32_t slot,
                           int rank, int world_size, const uint32_t *d_col_len, uint64_t nnz_global, float threshold,
                           int may_continue_prev, float back_threshold);
int gl_bfs_bits_shard_finish(gl_spmspv_plan csc, gl_spmv_plan rows, uint32_t *d_ctl, uint32_t *d_tally, const uint32_t *d_tally_in,
                             uint32_t last_slot, int rank, int world_size, uint64_t nnz_global, float threshold, int may_continue_last,
                             float back_threshold);
/* The same schedule as TWO launches per slot with the decisions fused into them, for a caller that meets the BFS in the
 * middle: the C++ module layer recognises the reference's pull iteration -- SpMV, eWiseAdd(+0), AssignVectorDense,
 * app/bfs.h:118-123 -- and runs it as gl_bfs_bits_push_step + gl_bfs_bits_pull_step on three rotating bit vectors
 * (include/graphlily/module/fusion.h).
 *   gl_bfs_bits_begin_from  control words as gl_bfs_bits_begin(first_pull_slot = 0) leaves them, d_bits = THREE vectors of
 *             bits_words words, the first = (x != 0), the other two cleared.  d_distance (may be NULL) is only read: with
 *             `rows` -- the whole-matrix GL_PLAN_BOOLEAN plan, which keeps the rows as CSR -- the non-zeros of the rows
 *             already reached are counted, so that a run starting in the middle of a BFS (app/bfs.h:195-216) goes bottom-up
 *             as early as one that ran from the source.
 *   gl_bfs_bits_push_step   the scattering push of slot `slot` when it pushes a light frontier (d_bits_out all zero on entry;
 *             d_bits_spare, if not NULL, is cleared for the next slot), or -- `rows` given -- the bottom-up scan.
 *   gl_bfs_bits_pull_step   the streaming pull; it follows the push step of its slot also when that one ran: it then only
 *             adds up the push step's totals and takes the slot's decisions. */
int gl_bfs_bits_begin_from(uint32_t *d_ctl, uint32_t ctl_words, const float *d_x, uint32_t n, uint32_t *d_bits, uint32_t bits_words,
                           const float *d_distance, gl_spmv_plan rows);
int gl_bfs_bits_push_step(gl_spmspv_plan csc, gl_spmv_plan rows, const uint32_t *d_bits_in, uint32_t *d_bits_out, uint32_t *d_bits_spare,
                          uint32_t bits_words, float *d_distance, float level, uint32_t *d_ctl, uint32_t slot, float threshold,
                          int may_continue);
int gl_bfs_bits_pull_step(gl_spmv_plan plan, gl_spmspv_plan csc, const uint32_t *d_bits_in, uint32_t *d_bits_out, float *d_distance,
                          float level, uint32_t *d_ctl, uint32_t slot, float threshold, int may_continue, float back_threshold);
/* Extension: direction switch inside the operator.  `pull` is an SpMV plan over the same matrix and row shard
 * (BFS holds both, app/bfs.h:83-99).  A run with zero == 0 whose frontier columns hold more than 1/8 of the matrix's
 * non-zeros (GRAPHLILY_SPMSPV_PULL_DIV; 0 = never) is then computed row-wise into the dense accumulator instead of being
 * binned -- at 24 bytes per product against 8 (4) per non-zero that is where the two cost the same: a GL_PLAN_BOOLEAN
 * plan serves (||,&&) (frontier -> bit vector -> boolean SpMV; results identical), a general / pattern plan serves (+,x)
 * (frontier -> dense vector -> SpMV; both ways add the float products in f64: same values up to that sum's order; not a
 * GL_PLAN_NO_MULADD plan) and (min,+) with zero <= FLOAT_INF (results identical: the operator's saturation at FLOAT_INF is
 * hidden by the final min with zero).  One of each kind may be attached; the decision is taken on the device.  The SpMV
 * plans are not owned and must outlive the attachment; NULL detaches both. */
int gl_spmspv_plan_attach_pull(gl_spmspv_plan plan, gl_spmv_plan pull);
/* Optional: an upper bound on the number of entries of the NEXT run's input vector (the drivers know it from
 * the previous iteration's count).  Frontiers too small to reach the threshold whatever their columns
 * are then skip the decision kernels.  One-shot: consumed by the next gl_spmspv_run. */
int gl_spmspv_plan_hint(gl_spmspv_plan plan, uint32_t vector_nnz_upper_bound);
/* One-shot: everything a module that has just uploaded the vector from the host knows about it -- its entries, the non-zeros
 * their columns hold in total (`work`), the LONGEST of those columns.  Up to 1024 entries and 2048 non-zeros the run is then
 * ONE launch of one workgroup (products by returning atomics, sort of the rows reached, ordered emission); a run whose work
 * stays below the direction switch's threshold skips the decision kernel and the row-wise kernels that would only find the
 * switch closed; and the bin launch is sized by the work.  Results never depend on the hint: the kernels check what they
 * find (a vector that is not tiny after all is computed correctly by the one workgroup, slowly; a wrong "light" makes the
 * run bin a vector it would have applied row-wise). */
int gl_spmspv_plan_hint_work(gl_spmspv_plan plan, uint32_t vector_nnz, uint64_t work, uint32_t longest_column);
/* which way the last run went (1 = row-wise).  Blocking; for tests and reports. */
int gl_spmspv_last_direction(gl_spmspv_plan plan, int *row_wise);

/* SpMSpVModule::get_results_nnz (module/spmspv_module.h:239-242): the one
 * device->host control read per push iteration.  Blocking. */
int gl_sparse_nnz(const gl_idx_val *d_sparse, uint32_t *nnz);

/* ---------------------------------------------------------------- apply ops */
/* mode 3, eWiseAddModule::run (module/add_scalar_vector_dense_module.h:179-192,
 * hw/kernel_add_scalar_vector_dense_impl.h:6-27): out[i] = in[i] + val. */
int gl_ewise_add(const float *d_in, float *d_out, uint32_t len, float val);

/* mode 4, AssignVectorDenseModule::run (module/assign_vector_dense_module.h:208-220,
 * hw/kernel_assign_vector_dense_impl.h:8-47): WriteToZero: mask[i]==0 -> inout[i]=val;
 * WriteToOne: mask[i]!=0 -> inout[i]=val; NOMASK is GL_ERR_INVALID_ARG. */
int gl_assign_dense(const float *d_mask, float *d_inout, uint32_t len, float val, int mask_type);

/* mode 5, AssignVectorSparseModule::run(val) (module/assign_vector_sparse_module.h:278-292,
 * hw/kernel_assign_vector_sparse_no_new_frontier_impl.h:4-55):
 * inout[mask[k].index] = val for k in 1..mask[0].index.  max_entries bounds the
 * launch (capacity of d_mask minus the head). */
int gl_assign_sparse(const gl_idx_val *d_mask, float *d_inout, float val, uint32_t max_entries);

/* mode 6, AssignVectorSparseModule::run() (module/assign_vector_sparse_module.h:295-303,
 * hw/kernel_assign_vector_sparse_new_frontier_impl.h:4-78): for every mask entry
 * (idx,v): if inout[idx] > v { inout[idx] = v; push (idx,v) }.  d_new_frontier
 * gets head {count, 0} and the pushed entries in mask order. d_scratch is optional. */
int gl_assign_sparse_new_frontier(const gl_idx_val *d_mask, float *d_inout,
                                  gl_idx_val *d_new_frontier, uint32_t max_entries);

/* ------------------------------------------------- the reference's other value types
 * graphlily/global.h:62-64: val_t is `unsigned`, ap_ufixed<32, 8, AP_RND, AP_SAT> (the shipped default) or float.
 * Both integer types are 32-bit words; they use the float entry points' buffers, plans (gl_*_plan_create take the
 * value words through the `float *` parameter, bit for bit -- create SpMV plans without GL_PLAN_BOOLEAN) and
 * sparse elements ({uint32 index; uint32 value}), and these entry points with the value type spelled out.
 * Semantics (hw/ufixed_pe_fwd.h:23-65):
 *   GL_VAL_UNSIGNED     + and * wrap mod 2^32, a && b / a || b give 1, MIN is unsigned; all three semirings
 *                       (modular sums are exact in any order).
 *   GL_VAL_UFIXED_32_8  value = word / 2^24; a + b saturates at 2^32 - 1 (AP_SAT), a && b / a || b give 1.0 = 1 << 24,
 *                       MIN is the unsigned minimum of the words; (||,&&) and (min,+).  (+,x) returns
 *                       GL_ERR_UNSUPPORTED: a saturating, rounding sum depends on the order of its terms.
 * Mask tests compare words with 0 (SpMV, dense assign) or with `zero_bits` (SpMSpV), masked-off SpMV rows are 0.
 * Results are bit-exact against oracle/graphlily_oracle.c's integer restatement (tests/test_gpu_typed.py). */
#define GL_VAL_FLOAT 0
#define GL_VAL_UNSIGNED 1
#define GL_VAL_UFIXED_32_8 2
int gl_spmv_run_typed(gl_spmv_plan plan, const void *d_x, const void *d_mask, void *d_y, int op, uint32_t zero_bits,
                      int mask_type, int val_type);
int gl_spmspv_run_typed(gl_spmspv_plan plan, const void *d_vector, const void *d_mask, void *d_result, int op,
                        uint32_t zero_bits, int mask_type, int val_type);
int gl_ewise_add_typed(const void *d_in, void *d_out, uint32_t len, uint32_t val_bits, int val_type);
int gl_assign_dense_typed(const void *d_mask, void *d_inout, uint32_t len, uint32_t val_bits, int mask_type, int val_type);
int gl_assign_sparse_typed(const void *d_mask, void *d_inout, uint32_t val_bits, uint32_t max_entries);
int gl_assign_sparse_new_frontier_typed(const void *d_mask, void *d_inout, void *d_new_frontier, uint32_t max_entries,
                                        int val_type);
int gl_sparse_to_dense_typed(const void *d_sparse, void *d_dense, uint32_t range, uint32_t zero_bits, uint32_t max_entries);

/* ------------------------------------------------------------------ multi-GPU exchange (RCCL over xGMI)
 * The reference is single-device; SURVEY.md 8(e): the matrix is cut into contiguous row ranges, one per GPU /
 * process (gl_*_plan_create's [row_begin,row_end), SpMVModule::set_row_shard), every iteration each rank produces its
 * slice of the result vector and ONE all-gather rebuilds the whole vector -- the next iteration's input -- on every
 * rank.  These calls are that exchange step for C / C++ callers (graphlily_amd/dist.py does the same through
 * torch.distributed): asynchronous, on the library's stream, in place.  RCCL is loaded at run time.
 *   gl_dist_unique_id   rank 0 obtains 128 bytes and hands them to the other ranks by any means (file, MPI, socket);
 *   gl_dist_init        every rank, after gl_init(its device): ncclCommInitRank;
 *   gl_dist_all_gather_f32     d_full[bounds[r] .. bounds[r+1]) is valid on rank r before, everywhere after;
 *                              slices may differ in length (nnz-balanced ranges): point-to-point pushes in one group;
 *   gl_dist_all_gather_bits_tally   the same for a bit vector (gl_pack_bits / gl_bfs_pull_step), bounds in ROWS, multiples
 *                              of 32 -- a sharded BFS exchanges n/8 bytes per iteration instead of 4n -- and, in the same
 *                              grouped operation, every rank's tallies of the slot (gl_bfs_bits_shard_step: d_tally_slot =
 *                              the slot's table, world_size blocks of bytes_per_rank = 4 * GL_BFS_TALLY_RANK_WORDS bytes,
 *                              rank r's block at r * bytes_per_rank; NULL / 0: the bits alone);
 *   gl_dist_all_gather_sparse  the ranks' sparse result lists (disjoint ascending row ranges) concatenated in rank order
 *                              into d_full with head {total, head_val}; reads the counts back (blocking), like
 *                              SpMSpVModule::get_results_nnz in the reference's push loops. */
typedef struct gl_dist_s *gl_dist;
/* gl_dist_slice_plan: the host arithmetic of the three all-gathers -- which BYTES of the exchanged buffer each rank owns:
 * GL_DIST_F32 (in: world + 1 element bounds), GL_DIST_BITS (in: world + 1 row bounds, multiples of 32 except the last;
 * whole words), GL_DIST_SPARSE (in: world entry counts; entries follow the head element in rank order).  Needs neither a
 * device nor RCCL, so the uneven-bounds cases are unit-tested on the CPU. */
#define GL_DIST_F32 0
#define GL_DIST_BITS 1
#define GL_DIST_SPARSE 2
int gl_dist_slice_plan(int kind, int world_size, const uint32_t *in, uint64_t *lo_bytes, uint64_t *hi_bytes);
int gl_dist_unique_id(void *id128);
int gl_dist_init(gl_dist *comm, int rank, int world_size, const void *id128);
int gl_dist_destroy(gl_dist comm);      /* GL_ERR_INVALID_ARG while a gl_graph that recorded one of its exchanges is alive */
int gl_dist_rank(gl_dist comm, int *rank, int *world_size);
int gl_dist_all_gather_f32(gl_dist comm, float *d_full, const uint32_t *bounds);
int gl_dist_all_gather_bits_tally(gl_dist comm, uint32_t *d_bits, const uint32_t *row_bounds, uint32_t *d_tally_slot, uint32_t bytes_per_rank);
int gl_dist_all_gather_sparse(gl_dist comm, const gl_idx_val *d_local, gl_idx_val *d_full, uint32_t capacity, float head_val,
                              uint32_t *total);

/* ---------------------------------------------------------------- utilities */
/* convert_sparse_vec_to_dense_vec (graphlily/global.h:153-164) on device; replaces
 * the host round trip at the push->pull switch (app/bfs.h:196-201). */
int gl_sparse_to_dense(const gl_idx_val *d_sparse, float *d_dense, uint32_t range, float zero,
                       uint32_t max_entries);

/* csr2csc (io/data_loader.h:108-144), host arrays in and out; rows inside a column stay ascending.  Output arrays:
 * csc_indptr[num_cols+1], csc_indices[nnz], csc_data[nnz].  Done on the GPU when the runtime is up and the matrix is large
 * (one stable radix sort of (row, value) pairs by column), on the host otherwise (parallel counting sort; needs no GPU):
 * identical output. */
int gl_csr2csc(uint32_t num_rows, uint32_t num_cols, const uint32_t *indptr, const uint32_t *indices,
               const float *data, uint32_t *csc_indptr, uint32_t *csc_indices, float *csc_data);
/* util_normalize_csr_matrix_by_outdegree (io/data_formatter.h:36-51): data[i] = 1.0 / (entries in the column of i),
 * double divide stored as float; host arrays, GPU when the runtime is up and the matrix is large. */
int gl_csr_normalize_by_outdegree(uint32_t num_rows, uint32_t num_cols, const uint32_t *indptr, const uint32_t *indices,
                                  float *data);

/* scipy-npz CSR loader: replaces cnpy::npz_load in
 * load_csr_matrix_from_float_npz (io/data_loader.h:51-70).  Two-call protocol:
 * gl_npz_csr_open parses the file and reports sizes, gl_npz_csr_read copies
 * into caller arrays and closes the handle (gl_npz_csr_close to abandon). */
typedef struct gl_npz_csr_s *gl_npz_csr;
int gl_npz_csr_open(const char *path, gl_npz_csr *handle,
                    uint32_t *num_rows, uint32_t *num_cols, uint64_t *nnz);
int gl_npz_csr_read(gl_npz_csr handle, float *data, uint32_t *indices, uint32_t *indptr);
int gl_npz_csr_close(gl_npz_csr handle);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHLILY_HIP_H_ */
