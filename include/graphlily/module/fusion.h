// graphlily/module/fusion.h -- what the module layer does BETWEEN the reference's module calls (no counterpart in the
// reference; everything here is invisible to a caller of the module API).
//
// 1. Pairing.  The reference's apps hold one matrix twice: BFS and SSSP own an SpMVModule (CSR) and an SpMSpVModule (CSC)
//    built from the same file (app/bfs.h:83-99, app/sssp.h:129-147).  Modules of ONE ModuleCollection whose plans have the
//    same shape and non-zero count are taken to be such a pair: the SpMSpV plan gets the SpMV plan attached
//    (gl_spmspv_plan_attach_pull: a push iteration whose frontier is heavy runs row-wise), and the pair is what a fused BFS
//    pull iteration needs (below).  GRAPHLILY_MODULE_FUSION=0 switches both 1. and 2. off.
//
// 2. The BFS pull iteration.  The reference's drivers express it as three module calls (app/bfs.h:118-123, :208-216):
//        SpMV->run()                       (||,&&), masked WriteToZero by the distances D:   results = mask(A x)
//        eWiseAdd->run(n, 0)               results -> vector                                  (the next frontier)
//        DenseAssign->run(n, iter + 1)     WriteToOne by vector:                              D[vector != 0] = iter + 1
//    On non-blocking modules (owned by a ModuleCollection) the first two are DEFERRED; when the third arrives with exactly
//    those bindings the three run as the two launches of the bit-frontier schedule (gl_bfs_bits_push_step +
//    gl_bfs_bits_pull_step on three rotating bit vectors: streaming pull, or -- late in a BFS -- the bottom-up scan of the
//    rows not reached yet), which write D and keep the frontier as BITS.  The float `vector` and `results` buffers are then
//    OWED: whoever touches one of them through DeviceBuffer::ptr() (a download, another module, the next non-matching call)
//    first has them written from the bits (gl_unpack_bits), so every observable value is the unfused sequence's.  Any other
//    call sequence simply runs the deferred calls as they are.  GRAPHLILY_MODULE_FUSION=0 switches it off.
//
// 3. The pull loops of SSSP and PageRank.  The reference's drivers express an iteration as two module calls:
//        SpMV->run()                 no mask:                         results = A x          (app/sssp.h:160-163, app/pagerank.h:85-87)
//        eWiseAdd->run(n, val)       in = results, out = vector:       vector = results + val (val = 0: a copy; PageRank: the teleport term)
//    On non-blocking modules the first is deferred; when the second arrives with exactly those bindings the SpMV runs with
//    `val` folded into its epilogue -- zero + sum + val is computed as (val) + sum, which is the same float when zero is 0 (the
//    only case taken for val != 0); for val == 0 nothing changes -- and the two buffers SWAP their device blocks instead of
//    being copied: `vector` holds what the copy would have left, 8 n bytes and one launch less per iteration.  `results` is
//    then OWED: whoever reads it before the next SpMV overwrites it gets it written first (val == 0: a copy of vector; else
//    the SpMV once more on the old vector, which the swap left in results' block).
#ifndef GRAPHLILY_MODULE_FUSION_H_
#define GRAPHLILY_MODULE_FUSION_H_

#include <cstdlib>
#include <functional>
#include <vector>

#include "graphlily/global.h"

namespace graphlily {
namespace module {
namespace detail {

struct PullFusion {
    // ---------------------------------------------------------------- pairing
    struct Entry {
        const void *module, *owner;
        gl_spmv_plan spmv;
        gl_spmspv_plan spmspv;
        uint32_t rows, cols;
        uint64_t nnz;
    };
    std::vector<Entry> entries;

    static bool env_on(const char *name) {
        const char *e = getenv(name);
        return !(e && atoi(e) == 0);
    }

    void forget(const void *module) {
        flush();
        if (last_copy_module == module) {           // its plan is about to change or go: what ITS swap left owed is written now
            if (last_copy_res.owed()) (void)last_copy_res.ptr();   // (the debt re-runs this module's SpMV: it cannot outlive it)
            last_copy_res = DeviceBuffer();
            last_copy_module = nullptr;
        }
        if (copy_module == module) {
            rerun_spmv = nullptr;
            copy_module = nullptr;
        }
        for (size_t i = 0; i < entries.size();)
            if (entries[i].module == module) entries.erase(entries.begin() + i);
            else i++;
        if (bits_module == module) {
            settle_last();
            bits_for_vec = nullptr;
        }
    }

    // a module has (re)built its plan; whole-matrix plans only
    void announce(const void *module, const void *owner, gl_spmv_plan spmv, gl_spmspv_plan spmspv, uint32_t rows, uint32_t cols,
                  uint64_t nnz) {
        forget(module);
        entries.push_back(Entry{module, owner, spmv, spmspv, rows, cols, nnz});
        if (!owner || !env_on("GRAPHLILY_MODULE_FUSION")) return;
        for (const Entry &a : entries)
            for (const Entry &b : entries)
                if (a.spmv && b.spmspv && a.owner == owner && b.owner == owner && a.rows == b.rows && a.cols == b.cols && a.nnz == b.nnz)
                    (void)gl_spmspv_plan_attach_pull(b.spmspv, a.spmv);   // (a plan of another layout just stays unattached)
    }

    gl_spmspv_plan partner_of(const void *spmv_module) const {
        for (const Entry &a : entries)
            if (a.module == spmv_module && a.spmv && a.owner)
                for (const Entry &b : entries)
                    if (b.spmspv && b.owner == a.owner && b.rows == a.rows && b.cols == a.cols && b.nnz == a.nnz) return b.spmspv;
        return nullptr;
    }

    // ---------------------------------------------------------------- the deferred pull iteration
    int stage = 0;                                  // 0 nothing pending, 1 SpMV deferred, 2 and eWiseAdd(+0) too
    std::function<void()> run_spmv, run_ewise;      // the deferred calls as they would have run
    DeviceBuffer vec, dist, res;                    // the SpMV's vector / mask (= distances) / results at the time of its run()
    gl_spmv_plan plan = nullptr;
    gl_spmspv_plan csc = nullptr;
    const void *spmv_module = nullptr;
    uint32_t n = 0;
    // the frontier as bits: control words + three vectors, valid for `bits_for_vec` while that buffer is still owed
    static constexpr uint32_t kCtlWords = 528;      // 16 + 2 x 256 slots
    DeviceBuffer bits;
    const void *bits_for_vec = nullptr, *bits_module = nullptr;
    uint64_t words = 0;
    uint32_t slot = 1, cur = 0;

    uint32_t *ctl() const { return (uint32_t *)bits.raw(); }
    uint32_t *vecbits(uint32_t k) const { return (uint32_t *)bits.raw() + kCtlWords + (size_t)k * words; }

    std::function<void()> vec_prev_debt;            // what the vector owed before stage 2 put its own hook on it

    // run whatever is deferred, as it is
    void flush() {
        if (!stage) return;
        if (copy_kind) {                            // a deferred no-mask SpMV that no matching eWiseAdd followed: as it is
            stage = 0;
            copy_kind = false;
            std::function<void(float, bool)> r;
            r.swap(run_copy_spmv);
            res.settle_quietly();
            vec = res = DeviceBuffer();
            if (r) r(0.0f, false);
            return;
        }
        const int st = stage;
        stage = 0;
        res.settle_quietly();
        if (st == 2) {                              // the vector gets its earlier debt back: the SpMV below reads it
            vec.settle_quietly();
            std::function<void()> o;
            o.swap(vec_prev_debt);
            if (o) vec.owe(o);
        }
        std::function<void()> a, b;
        a.swap(run_spmv);
        b.swap(run_ewise);
        DeviceBuffer keep_v = vec, keep_d = dist, keep_r = res;   // (the calls below may re-enter defer_spmv)
        vec = dist = res = DeviceBuffer();
        if (a) a();
        if (st == 2 && b) b();
    }

    // SpMVModule::run on a non-blocking module with the (||,&&) semiring, masked WriteToZero, on an unsplit whole-matrix
    // boolean plan that has a partner: do not launch yet
    void defer_spmv(const void *module, gl_spmv_plan p, gl_spmspv_plan c, uint32_t n_, DeviceBuffer vector, DeviceBuffer mask,
                    DeviceBuffer results, std::function<void()> run_now) {
        flush();
        stage = 1;
        spmv_module = module;
        plan = p;
        csc = c;
        n = n_;
        vec = vector;
        dist = mask;
        res = results;
        run_spmv = std::move(run_now);
        res.owe([this] { flush(); });               // a reader of the results gets them
    }

    // eWiseAddModule::run: does it continue the chain?
    bool defer_ewise(const DeviceBuffer &in, const DeviceBuffer &out, uint32_t len, float val, std::function<void()> run_now) {
        if (stage == 1 && copy_kind) return complete_copy(in, out, len, val);
        if (stage != 1 || in.id() != res.id() || out.id() != vec.id() || len != n || val != 0.0f) return false;
        stage = 2;
        run_ewise = std::move(run_now);
        // from here on a reader of the vector expects the NEW frontier: it gets whatever the vector owed before (the float
        // values a fused iteration kept as bits) and then the deferred calls
        vec_prev_debt = nullptr;
        if (vec.owed()) {
            DeviceBuffer v = vec;
            // (take the old debt off the handle without running it)
            vec_prev_debt = take_debt(v);
        }
        vec.owe([this] {
            std::function<void()> o;
            o.swap(vec_prev_debt);
            if (o) o();
            flush();
        });
        return true;
    }

    static std::function<void()> take_debt(const DeviceBuffer &b) { return b.take_debt(); }

    // ---------------------------------------------------------------- SpMV + eWiseAdd(n, val) of the SSSP / PageRank pull loops
    bool copy_kind = false;
    bool copy_zero_is_0 = false, copy_muladd = false;
    std::function<void(float, bool)> run_copy_spmv;   // (extra term, fold it into the epilogue?) -> the SpMV on the bound buffers
    std::function<void(const DeviceBuffer &, const DeviceBuffer &)> rerun_spmv;   // (x, y): the plain SpMV on explicit buffers
    DeviceBuffer last_copy_res;                     // the results buffer the last swap left owed ...
    const void *last_copy_module = nullptr;         // ... and the module whose SpMV that debt would re-run
    const void *copy_module = nullptr;

    // SpMVModule::run on a non-blocking module, no mask, whole square matrix, general / pattern layout: do not launch yet
    void defer_copy_spmv(const void *module, uint32_t n_, DeviceBuffer vector, DeviceBuffer results, bool zero_is_0, bool muladd,
                         std::function<void(float, bool)> run_now, std::function<void(const DeviceBuffer &, const DeviceBuffer &)> rerun) {
        flush();
        stage = 1;
        copy_kind = true;
        copy_module = module;
        n = n_;
        vec = vector;
        res = results;
        copy_zero_is_0 = zero_is_0;
        copy_muladd = muladd;
        run_copy_spmv = std::move(run_now);
        rerun_spmv = std::move(rerun);
        res.owe([this] { flush(); });               // a reader of the results gets them
    }

    bool complete_copy(const DeviceBuffer &in, const DeviceBuffer &out, uint32_t len, float val) {
        const bool fold = val != 0.0f;
        if (in.id() != res.id() || out.id() != vec.id() || len != n || vec.size() != res.size() || (fold && !(copy_muladd && copy_zero_is_0)))
            return false;                           // (the caller's barrier_() runs the deferred SpMV as it is)
        stage = 0;
        copy_kind = false;
        res.settle_quietly();
        (void)vec.rptr();                           // (whatever the vector itself still owed)
        if (last_copy_res.valid() && last_copy_res.id() != res.id() && last_copy_res.owed()) (void)last_copy_res.ptr();
        vec.clear_on_write();                       // (the previous swap's hook: `results` is about to be overwritten, or was just settled)
        std::function<void(float, bool)> r;
        r.swap(run_copy_spmv);
        r(val, fold);                               // results' block = A x (+ val)
        vec.swap_storage(res);                      // `vector` now names it; `results` names the old x
        DeviceBuffer v = vec, rr = res;
        std::function<void(const DeviceBuffer &, const DeviceBuffer &)> again = rerun_spmv;
        if (!fold) {
            rr.owe([v, rr] {                        // results = vector (what the copy would have left in both)
                GRAPHLILY_CHECK(gl_buf_d2d(rr.raw(), v.raw(), rr.size()));
                v.clear_on_write();
            });
            // ... taken when somebody reads `results` -- or just before anybody gets to WRITE `vector` (an upload, another module's
            // output bound to it): the copy must be of A x, not of what comes next.  The loop's own SpMV reads through rptr().
            v.on_write([rr] { if (rr.owed()) (void)rr.ptr(); });
        } else {
            rr.owe([rr, again] {                    // results = A x_old, and x_old is what results' block holds
                DeviceBuffer tmp(rr.size());
                again(rr, tmp);
                GRAPHLILY_CHECK(gl_buf_d2d(rr.raw(), tmp.raw(), rr.size()));
            });
        }
        last_copy_res = rr;
        last_copy_module = copy_module;
        vec = res = DeviceBuffer();
        return true;
    }

    // AssignVectorDenseModule::run: does it complete the chain?  Then run the three calls as the fused step.
    bool fire(const DeviceBuffer &mask, const DeviceBuffer &inout, uint32_t len, float val, int mask_type) {
        if (stage != 2 || mask.id() != vec.id() || inout.id() != dist.id() || len != n || mask_type != (int)kMaskWriteToOne) return false;
        if (!bits.valid() || bits_module != spmv_module) {
            uint64_t w = 0;
            GRAPHLILY_CHECK(gl_spmv_plan_bits_words(plan, &w));
            words = (w + 3u) & ~(uint64_t)3u;
            bits = DeviceBuffer(sizeof(uint32_t) * ((size_t)kCtlWords + 3u * words));
            bits_module = spmv_module;
            bits_for_vec = nullptr;
        }
        // the hooks of the deferred stages come off; what the vector owed BEFORE them decides where the frontier is
        res.settle_quietly();
        vec.settle_quietly();
        std::function<void()> old;
        old.swap(vec_prev_debt);
        const bool have_bits = bits_for_vec == vec.id() && (bool)old && slot + 2u < (kCtlWords - 16u) / 2u;
        if (!have_bits) {
            // the frontier comes as the float vector (first iteration, somebody has touched it since, or the slots ran out)
            if (old) old();                          // (the float values an earlier fused iteration still owed)
            settle_last();                           // (... and those of another vector, before its bits are overwritten)
            GRAPHLILY_CHECK(gl_bfs_bits_begin_from(ctl(), kCtlWords, (const float *)vec.raw(), n, vecbits(0), (uint32_t)words,
                                                   (const float *)dist.ptr(), plan));
            slot = 1;
            cur = 0;
        }
        // buffers an EARLIER fused step left owed that are not this step's (a driver that rebinds its results or vector between
        // iterations): their values are in the bit vector `cur`, which this step is about to rotate away from -- write them now
        if (last_vec.valid() && last_vec.id() != vec.id() && last_vec.owed()) (void)last_vec.ptr();
        if (last_res.valid() && last_res.id() != res.id() && last_res.owed()) (void)last_res.ptr();
        const uint32_t nxt = (cur + 1u) % 3u, spare = (cur + 2u) % 3u;
        float *d = (float *)dist.ptr();
        int rc = gl_bfs_bits_push_step(csc, plan, vecbits(cur), vecbits(nxt), vecbits(spare), (uint32_t)words, d, val, ctl(), slot, -1.0f, 2);
        if (rc == GL_OK)
            rc = gl_bfs_bits_pull_step(plan, csc, vecbits(cur), vecbits(nxt), d, val, ctl(), slot, -1.0f, 2, 0.0f);
        if (rc != GL_OK) {
            // this plan cannot run the fused step after all: the three calls as they are, now and from now on
            if (have_bits) GRAPHLILY_CHECK(gl_unpack_bits(vecbits(cur), n, (float *)vec.raw()));
            bits_for_vec = nullptr;
            fusable_ = false;
            stage = 1;                               // (the vector carries no hook any more: flush() as from stage 1, then the copy)
            std::function<void()> e;
            e.swap(run_ewise);
            flush();
            if (e) e();
            return false;
        }
        cur = nxt;
        slot++;
        bits_for_vec = vec.id();
        // both float buffers now hold the PREVIOUS iteration's values: owed
        DeviceBuffer v = vec, r = res;
        const uint32_t nn = n, at = cur;            // (the bit vector these two buffers' values are in: pinned, not "the current one")
        std::function<void()> settle = [this, v, r, nn, at] {
            v.settle_quietly();
            r.settle_quietly();
            const uint32_t *b = vecbits(at);
            GRAPHLILY_CHECK(gl_unpack_bits(b, nn, (float *)v.raw()));
            GRAPHLILY_CHECK(gl_unpack_bits(b, nn, (float *)r.raw()));
            bits_for_vec = nullptr;                 // whoever touched the vector may change it: pack again next time
        };
        vec.owe(settle);
        res.owe(settle);
        last_vec = vec;
        last_res = res;
        dist.mark_levels(val);                      // (the distances now hold levels up to `val`, as far as this layer knows)
        stage = 0;
        run_spmv = nullptr;
        run_ewise = nullptr;
        vec = dist = res = DeviceBuffer();
        return true;
    }

    DeviceBuffer last_vec, last_res;                // the buffers the last fused step left owed

    // the buffers an earlier fused step left owed are settled now (their bits are about to be overwritten, or to go away)
    void settle_last() {
        if (last_vec.owed()) (void)last_vec.ptr();
        if (last_res.owed()) (void)last_res.ptr();
        last_vec = last_res = DeviceBuffer();
    }

    bool fusable_ = true;
    bool enabled() const { return fusable_ && env_on("GRAPHLILY_MODULE_FUSION"); }
};

inline PullFusion &fusion() {
    static PullFusion *f = new PullFusion();   // never destroyed: modules may outlive static destruction order
    return *f;
}

}  // namespace detail
}  // namespace module
}  // namespace graphlily

#endif  // GRAPHLILY_MODULE_FUSION_H_
