#!/usr/bin/env python
"""Generates tests/golden/reference_test_cases.json: the cases of the reference's own module tests whose inputs are
fully determined by the test source, with expected outputs derived INDEPENDENTLY of oracle/ and of the HIP library.

Which cases (reference file:line), how the inputs are rebuilt, and how each expectation is derived:

  tests/test_module_apply.cpp:55-75    eWiseAdd            128 values rand()%10/100, val 1
  tests/test_module_apply.cpp:78-103   AssignVectorDense   mask / inout rand()%2, val 23, kMaskWriteToOne
  tests/test_module_apply.cpp:106-143  AssignVectorSparse  (no new frontier) 819 mask entries at stride 10 into 8192, val 3
  tests/test_module_apply.cpp:146-206  AssignVectorSparse  (new frontier, the SSSP relax) 12 entries at stride 10 into
                                                            128 values that are 5 or inf (= UFIXED_INF = 255 for the shipped val_t)
  tests/test_module_spmv_spmspv.cpp:139-166  SpMV on dense_32 (every entry of a 32 x 32 block present, padded to
                                                            128 x 128, values 1/128): 2 semirings x 3 masks, x / mask rand()%2
  tests/test_module_spmv_spmspv.cpp:262-282  SpMSpV on the "bank conflict" matrix built in the test source itself
                                                            (1024 x 1024, column i holds rows j*8 + i%8), dense vector
  tests/test_module_spmv_spmspv.cpp:296-313  SpMSpV on dense1K (every entry of 1024 x 1024 present, values 1/1024),
                                                            vector sparsity 0.99: 3 semirings x 3 masks

Inputs.  The reference tests never seed rand(), so every run of a test binary draws the same numbers from glibc's
default-seeded generator.  This script draws them from the same generator (ctypes -> libc srand(1); rand()) in the
order the test source does, restarting at the top of each test file (the tests before these in each file draw nothing).
dense_32 / dense1K are described by their names and use (all entries present); their .npz files are not needed.

Expectations come from closed forms written down here, not from a loop over the matrix:
  * dense matrices: every row sees the same terms, so y[r] is one number: (+,x) the float32 sum of a_i * v_i taken in
    ascending i (the order of the reference loops, spmv_module.h:489-497 / spmspv_module.h:466-486), evaluated with
    numpy float32 scalars; (||,&&) "any v_i != 0" (a = 1/n != 0); (min,+) min(zero, min_i(a + v_i)) in float32.
  * conflict matrix: row r is hit by exactly the columns i with i % 8 == r % 8, in ascending i.
  * masks: SpMV writes literal 0 where the mask forbids (spmv_module.h:518-530); SpMSpV writes semiring.zero there
    and compares the mask with semiring.zero (spmspv_module.h:499-516).
  * element-wise modules: numpy boolean indexing.
The JSON stores float32 values as exact decimal repr of the float32 (json round-trips them through float64 exactly).
"""
import ctypes
import json
import os

import numpy as np

F = np.float32
UFIXED_INF = F(255.0)   # graphlily/global.h:79: what "inf" and TropicalSemiring.zero are for the shipped val_t (:63, :100)

libc = ctypes.CDLL("libc.so.6")
libc.rand.restype = ctypes.c_int


def restart():
    libc.srand(1)   # the state a process starts with


def rand():
    return libc.rand()


def fl(a):
    return [float(F(v)) for v in a]


def seq_sum_f32(terms):
    """float32 sum in the given order (what `acc += a * b` does in a float accumulator)"""
    acc = F(0)
    for t in terms:
        acc = F(acc + F(t))
    return acc


def apply_cases():
    restart()
    out = {}
    # ---- eWiseAdd (:55-75)
    n = 128
    inp = np.array([F(rand() % 10) / F(100) for _ in range(n)], dtype=F)
    out["ewise_add"] = {"ref": "tests/test_module_apply.cpp:55-75", "in": fl(inp), "val": 1.0,
                        "derivation": "out[i] = float32(in[i] + 1)", "expected": fl(inp + F(1))}
    # ---- AssignVectorDense (:78-103): mask first, then inout
    mask = np.array([F(rand() % 2) for _ in range(n)], dtype=F)
    inout = np.array([F(rand() % 2) for _ in range(n)], dtype=F)
    exp = inout.copy()
    exp[mask != 0] = F(23)
    out["assign_dense_write_to_one"] = {"ref": "tests/test_module_apply.cpp:78-103", "mask": fl(mask), "inout": fl(inout), "val": 23.0,
                                        "mask_type": "kMaskWriteToOne", "derivation": "inout[mask != 0] = 23", "expected": fl(exp)}
    # ---- AssignVectorSparse, no new frontier (:106-143)
    size, val = 8192, F(3)
    length = int(np.floor(size * (1 - F(0.9))))           # (unsigned)floor(inout_size * (1 - mask_sparsity)) in float
    inc = size // length
    mvals = [F(rand() % 10) for _ in range(length)]
    midx = [i * inc for i in range(length)]
    inout = np.array([F(rand() % 10) for _ in range(size)], dtype=F)
    exp = inout.copy()
    exp[np.array(midx)] = val
    out["assign_sparse"] = {"ref": "tests/test_module_apply.cpp:106-143", "length": length, "increment": inc, "mask_index": midx,
                            "mask_val": fl(mvals), "inout": fl(inout), "val": 3.0,
                            "derivation": "inout[mask_index] = 3 (mask values are ignored)", "expected": fl(exp)}
    # ---- AssignVectorSparse, new frontier (:146-206)
    size = 128
    length = int(np.floor(size * (1 - F(0.9))))
    inc = size // length
    mvals = np.array([F(rand() % 10) for _ in range(length)], dtype=F)
    midx = np.array([i * inc for i in range(length)])
    inout = np.array([F(5) if (rand() % 10) > 5 else UFIXED_INF for _ in range(size)], dtype=F)
    relax = inout[midx] > mvals                            # indices are distinct, so entries do not interact
    exp = inout.copy()
    exp[midx[relax]] = mvals[relax]
    out["assign_sparse_new_frontier"] = {
        "ref": "tests/test_module_apply.cpp:146-206", "length": length, "increment": inc, "mask_index": midx.tolist(), "mask_val": fl(mvals),
        "inout": fl(inout), "inf": float(UFIXED_INF),
        "derivation": "mask indices are distinct: entry k relaxes iff inout[idx_k] > val_k; then inout[idx_k] = val_k and "
                      "{idx_k, val_k} joins the new frontier in mask order; head = {count, 0}",
        "expected_inout": fl(exp), "expected_frontier_index": midx[relax].tolist(), "expected_frontier_val": fl(mvals[relax])}
    return out


def spmv_cases():
    restart()
    n_live, n = 32, 128            # dense_32 rounded up to num_hbm_channels * pack_size = 128 (test :148-151)
    a = F(1.0 / n)                 # `x = 1.0 / num_rows` after the rounding (:152)
    cases = []
    for skip_empty_rows in (False, True):      # the flag changes the FPGA layout only (:154-166)
        for mask_type in ("kNoMask", "kMaskWriteToZero", "kMaskWriteToOne"):
            for sem in ("Arithmetic", "Logical"):
                x = np.array([F(rand() % 2) for _ in range(n)], dtype=F)
                mask = np.array([F(rand() % 2) for _ in range(n)], dtype=F)
                if sem == "Arithmetic":
                    live = seq_sum_f32([F(a * x[c]) for c in range(n_live)])
                else:
                    live = F(1) if np.any(x[:n_live] != 0) else F(0)
                y = np.zeros(n, dtype=F)       # rows 32..127 are empty: accumulator stays semiring.zero = 0
                y[:n_live] = live
                if mask_type == "kMaskWriteToZero":
                    y[mask != 0] = 0
                elif mask_type == "kMaskWriteToOne":
                    y[mask == 0] = 0
                cases.append({"semiring": sem, "mask_type": mask_type, "skip_empty_rows": skip_empty_rows, "x": fl(x), "mask": fl(mask),
                              "expected": fl(y)})
    # the reference calls them in the order (Arith, Logical) x NoMask, then x WriteToZero, then x WriteToOne: same as above
    return {"ref": "tests/test_module_spmv_spmspv.cpp:139-166", "matrix": "dense_32 padded to 128 x 128", "live": n_live, "n": n,
            "value": float(a), "derivation": "rows 0..31: (+,x) float32 sum over c < 32 of (1/128) * x[c] in ascending c; (||,&&) any(x[:32]); "
                                             "rows 32..127 empty -> 0; masked-off rows literal 0", "cases": cases}


def spmspv_vector(num_cols, sparsity):
    nnz = int(np.floor(num_cols * (1 - F(sparsity))))
    inc = num_cols // nnz
    vals = [F(rand() % 10) / F(10) for _ in range(nnz)]
    return [i * inc for i in range(nnz)], vals


def spmspv_cases():
    restart()
    out = {"ref": "tests/test_module_spmv_spmspv.cpp:262-313"}
    # ---- conflict1024, Arithmetic, NoMask, sparsity 0.00 (:283-284)
    n = 1024
    a = F(1.0 / n)
    idx, vals = spmspv_vector(n, 0.0)
    mask = [F(rand() % 2) for _ in range(n)]   # drawn although unused (kNoMask)
    y = np.zeros(n, dtype=F)
    for r in range(n):
        y[r] = seq_sum_f32([F(a * vals[i]) for i in range(r % 8, n, 8)])   # columns i with i % 8 == r % 8, ascending
    out["conflict1024"] = {"n": n, "value": float(a), "semiring": "Arithmetic", "mask_type": "kNoMask", "vector_index": idx,
                           "vector_val": fl(vals), "mask": fl(mask),
                           "derivation": "column i holds rows j*8 + i%8 (j < 128): row r receives (1/1024) * v_i from every column with "
                                         "i % 8 == r % 8, summed in float32 in ascending i", "expected": fl(y)}
    # ---- the cases on matrices this image does not have draw from rand() too: dense1K 0.50 (512 + 1024 draws),
    #      uniform10K10 0.50 (10000 columns and rows: 5000 + 10000), google+ 0.99 (:285-291).  google+'s dimension is not
    #      in the test source, so the generator state after it is unknown; the dense1K block below therefore restarts the
    #      generator instead of claiming the exact numbers the reference binary would see there.
    restart()
    zero_of = {"Arithmetic": F(0), "Logical": F(0), "Tropical": UFIXED_INF}   # global.h:96-100 as shipped
    cases = []
    for mask_type in ("kNoMask", "kMaskWriteToZero", "kMaskWriteToOne"):
        for sem in ("Arithmetic", "Logical", "Tropical"):
            idx, vals = spmspv_vector(n, 0.99)
            mask = np.array([F(rand() % 2) for _ in range(n)], dtype=F)
            zero = zero_of[sem]
            if sem == "Arithmetic":
                live = seq_sum_f32([F(a * v) for v in vals])
            elif sem == "Logical":
                live = F(1) if any(v != 0 for v in vals) else F(0)
            else:
                live = min([zero] + [F(a + v) for v in vals])
            y = np.full(n, live, dtype=F)
            if mask_type == "kMaskWriteToZero":
                y[mask != zero] = zero
            elif mask_type == "kMaskWriteToOne":
                y[mask == zero] = zero
            cases.append({"semiring": sem, "zero": float(zero), "mask_type": mask_type, "vector_index": idx, "vector_val": fl(vals),
                          "mask": fl(mask), "expected": fl(y)})
    out["dense1K"] = {"n": n, "value": float(a), "sparsity": 0.99,
                      "derivation": "every row sees all 10 vector entries: (+,x) float32 sum of (1/1024) * v_i in vector order; (||,&&) any(v_i != 0); "
                                    "(min,+) min(zero, min_i(1/1024 + v_i)); masked-off rows = semiring.zero, mask compared with semiring.zero "
                                    "(so the Tropical masks, 0/1 against 255, allow every row under WriteToOne and none under WriteToZero)",
                      "cases": cases}
    return out


def main():
    doc = {"_provenance": __doc__, "apply": apply_cases(), "spmv_dense_32": spmv_cases(), "spmspv": spmspv_cases()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_test_cases.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=None, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
