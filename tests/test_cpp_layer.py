"""The C++ side of the drop-in boundary: include/graphlily/ (header-only module layer over the C ABI).

CPU: the parity driver compiles with plain g++ -std=c++11, links the HIP library and fails loudly without
a GPU; where /root/reference exists, the reference's own app drivers compile UNMODIFIED against these
headers (oracle/Makefile `ref_apps`).
GPU: the parity driver runs every module against its compute_reference_results; the prebuilt reference
drivers (if they travelled with the snapshot) run BFS / PageRank / SSSP end to end on the HIP backend."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "graphlily_amd", "lib")
DRIVER = os.path.join(ROOT, "build", "modules_driver")
REF_APPS = os.path.join(ROOT, "oracle", "_ref", "ref_apps_on_hip")
APPS_DRIVER = os.path.join(ROOT, "build", "apps_driver")


def _build_apps_driver():
    """tests/cpp/apps_driver.cpp against include/graphlily/app/{bfs,sssp,pagerank}.h -- this repo's own app drivers (the
    reference's classes over the device-resident schedules); needs no reference tree."""
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "apps_driver.cpp"), "-o", APPS_DRIVER,
                           "-L", LIBDIR, "-lgraphlily_hip", "-Wl,-rpath," + LIBDIR])


def _build_driver():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "modules_driver.cpp"), "-o", DRIVER,
                           "-L", LIBDIR, "-lgraphlily_hip", "-Wl,-rpath," + LIBDIR])


def _build_typed_driver(define):
    """tests/cpp/typed_modules_driver.cpp with val_t = the reference's default fixed point (-DGRAPHLILY_VAL_UFIXED) or
    unsigned (-DGRAPHLILY_VAL_UNSIGNED): global.h:62-64 of the reference."""
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    out = os.path.join(ROOT, "build", "typed_modules_driver_" + define)
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-DGRAPHLILY_VAL_" + define,
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "typed_modules_driver.cpp"),
                           "-o", out, "-L", LIBDIR, "-lgraphlily_hip", "-Wl,-rpath," + LIBDIR])
    return out


@pytest.mark.parametrize("define", ["UFIXED", "UNSIGNED"])
def test_module_headers_instantiate_with_the_integer_value_types(define, tmp_path):
    from graphlily_amd import capi
    exe = _build_typed_driver(define)
    if capi.device_count() == 0:
        r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
        assert r.returncode != 0 and "gl_init" in r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/graphlily/app"), reason="reference tree not present")
@pytest.mark.parametrize("define", ["UFIXED", "UNSIGNED"])
def test_reference_app_drivers_compile_with_the_integer_value_types(define, tmp_path):
    """app/{bfs,pagerank,sssp}.h of the reference, unmodified, with val_t = its shipped default: compile + link only."""
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-DGRAPHLILY_VAL_" + define, "-I", os.path.join(ROOT, "include"),
                           "-I", "/root/reference", os.path.join(ROOT, "tests", "cpp", "ref_apps_compile.cpp"),
                           "-o", str(tmp_path / "ref_apps"), "-L", LIBDIR, "-lgraphlily_hip", "-Wl,-rpath," + LIBDIR])


def test_module_headers_compile_and_fail_loudly_without_gpu():
    from graphlily_amd import capi
    _build_driver()
    if capi.device_count() == 0:
        r = subprocess.run([DRIVER], capture_output=True, text=True)
        assert r.returncode != 0
        assert "gl_init" in r.stdout + r.stderr       # print-and-exit convention of the reference


def test_app_headers_compile_and_fail_loudly_without_gpu(tmp_path):
    from graphlily_amd import capi
    _build_apps_driver()
    if capi.device_count() == 0:
        r = subprocess.run([APPS_DRIVER, str(tmp_path / "none.npz"), str(tmp_path), "4"], capture_output=True, text=True)
        assert r.returncode != 0
        assert "gl_init" in r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/benchmark"), reason="reference tree not present")
@pytest.mark.parametrize("define", ["", "-DGRAPHLILY_USE_REFERENCE_APPS", "-DGRAPHLILY_VAL_UFIXED"])
def test_reference_bench_drivers_compile_against_the_app_headers(define, tmp_path):
    """benchmark/bench_{bfs,pagerank,sssp}.cpp of the reference, unmodified, with -I<repo>/include first: "graphlily/app/*.h"
    resolves to this repo's drivers; -DGRAPHLILY_USE_REFERENCE_APPS makes them step aside for the checkout's (#include_next)."""
    for app in ("bfs", "pagerank", "sssp"):
        subprocess.check_call(["g++", "-std=c++11", "-O1", "-w"] + ([define] if define else []) +
                              ["-I", os.path.join(ROOT, "include"), "-I", "/root/reference", "/root/reference/benchmark/bench_%s.cpp" % app,
                               "-o", str(tmp_path / ("bench_" + app)), "-L", LIBDIR, "-lgraphlily_hip", "-Wl,-rpath," + LIBDIR])
    out = subprocess.run(["nm", "-C", str(tmp_path / "bench_bfs")], capture_output=True, text=True).stdout
    # which BFS got compiled in: the device-resident schedule serves val_t = float only (dead code otherwise)
    assert ("gl_bfs_bits_shard_step" in out) == (define == "")


@pytest.mark.skipif(not os.path.isdir("/root/reference/benchmark"), reason="reference tree not present")
def test_reference_bench_spmspv_compiles_unmodified_and_its_build_mode_runs(tmp_path):
    """benchmark/bench_spmspv.cpp includes graphlily/synthesizer/overlay_synthesizer.h for `bench_spmspv <target> build`
    (:294-306) and counts on <cmath> / <fstream> arriving through the reference's headers: include/graphlily/synthesizer/ keeps
    the class's surface (the kernels are prebuilt HIP: nothing to synthesise).  The build mode needs no GPU."""
    exe = str(tmp_path / "bench_spmspv")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I", os.path.join(ROOT, "include"), "-I", "/root/reference",
                           "/root/reference/benchmark/bench_spmspv.cpp", "-o", exe, "-L", LIBDIR, "-lgraphlily_hip", "-Wl,-rpath," + LIBDIR])
    r = subprocess.run([exe, "hw", "build"], capture_output=True, text=True, cwd=str(tmp_path), timeout=60)
    assert r.returncode == 0 and "nothing to synthesise" in r.stdout and "Kernel Build Complete" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests"), reason="reference tree not present")
@pytest.mark.parametrize("define", ["", "-DGRAPHLILY_VAL_UFIXED"])
@pytest.mark.parametrize("suite", ["test_module_apply", "test_module_spmv_spmspv", "test_app"])
def test_reference_test_suites_compile_unmodified(suite, define, tmp_path):
    """The reference's own acceptance suites (tests/test_module_*.cpp and tests/test_app.cpp are callers of the
    boundary) compile UNMODIFIED against include/: <ap_fixed.h> = include/ap_fixed.h (the one type the reference takes from it),
    <gtest/gtest.h> = tests/cpp/gtest/gtest.h (test infrastructure), the cl::Buffer / cl::CommandQueue that
    test_module_apply.cpp:236-256 makes itself = include/graphlily/cl_buffers.h.  With no GPU every TEST stops at
    set_up_runtime's print-and-exit (xcl2.hpp:40-46) -- never a CPU path."""
    exe = str(tmp_path / suite)
    cmd = ["g++", "-std=c++11", "-O2", "-w"] + ([define] if define else []) + ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp"),
          "/root/reference/tests/%s.cpp" % suite, "-o", exe, "-L", LIBDIR, "-lgraphlily_hip", "-Wl,-rpath," + LIBDIR]
    subprocess.check_call(cmd)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "--gtest_filter=-Synthesize.*:Clean*"], capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode != 0 and "Error calling" in r.stdout and "[       OK ]" not in r.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/graphlily/app"), reason="reference tree not present")
def test_reference_app_drivers_compile_unmodified():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_apps"])
    assert os.path.exists(REF_APPS)


@pytest.mark.gpu
def test_cpp_module_layer_parity(gpu):
    _build_driver()
    r = subprocess.run([DRIVER], capture_output=True, text=True, timeout=90)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ALL CHECKS PASSED" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("n,nnz,iters", [(30000, 400000, 8), (600000, 7000000, 9)])
def test_cpp_app_drivers_match_the_oracle(gpu, tmp_path, n, nnz, iters):
    """include/graphlily/app/{bfs,sssp,pagerank}.h driven from C++ (tests/cpp/apps_driver.cpp): BFS pull / pull_push on the
    device-resident schedule (first call enqueued, then the replayed hipGraph; 600 K vertices: the packed read-back), push, the
    same through a world-of-one gl_dist_* communicator (the sharded schedule's path), SSSP in its three modes, PageRank.  Every
    result word for word against the oracle (BFS / SSSP) or to 1e-4 (PageRank, the reference tests' bar)."""
    import scipy.sparse as sp
    from graphlily_amd import datasets
    from oracle import oracle as O
    from helpers import to_oracle
    _build_apps_driver()
    m = datasets.rmat(n, nnz, seed=3)
    A = sp.csr_matrix((m.adj_data, m.adj_indices.astype(np.int32), m.adj_indptr.astype(np.int32)), shape=(m.num_rows, m.num_cols),
                      dtype=np.float32)
    p = str(tmp_path / "rmat_csr_float32.npz")
    sp.save_npz(p, A, compressed=False)
    r = subprocess.run([APPS_DRIVER, p, str(tmp_path), str(iters), "dist"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(" OK") == 17, r.stdout[-3000:]      # (12 + the zero-iteration first call and the four alternating-width pulls)

    def rd(name):
        return np.fromfile(str(tmp_path / (name + ".bin")), dtype=np.float32)

    ob = to_oracle(m)
    O.util_round_csr_matrix_dim(ob, 128, 128)
    ob.adj_data = np.ones(ob.nnz, np.float32)
    ref = O.bfs(ob, 0, iters)
    for name in ("bfs_pull_push", "bfs_pull", "bfs_push", "bfs_pull_push_dist"):
        assert np.array_equal(rd(name).view(np.uint32), ref.view(np.uint32)), name
    assert len(np.unique(ref)) > 3
    os_ = to_oracle(m)
    O.sssp_preprocess(os_)
    O.util_round_csr_matrix_dim(os_, 128, 128)
    ref = O.sssp(os_, 0, iters, zero=999999999.0)
    for name in ("sssp_pull_push", "sssp_pull", "sssp_push"):
        assert np.array_equal(rd(name).view(np.uint32), ref.view(np.uint32)), name
    op = to_oracle(m)
    O.util_round_csr_matrix_dim(op, 128, 128)
    O.util_normalize_csr_matrix_by_outdegree(op)
    op.adj_data = (op.adj_data * np.float32(0.9)).astype(np.float32)
    ref = O.pagerank(op, 0.9, 10)
    assert np.abs(rd("pagerank").astype(np.float64) - ref).max() <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("define", ["UFIXED", "UNSIGNED"])
def test_cpp_module_layer_integer_value_types(gpu, define, tmp_path):
    """Every module instantiated with val_t = ap_ufixed<32,8> / unsigned, run on the GPU by the C++ driver; each output must
    equal the oracle's integer restatement on the inputs the driver wrote, word for word.  Also pins the HOST conversions of
    graphlily::ufixed_32_8 / value_kind<>::from_float against the oracle's (io/data_loader.h:75-90)."""
    from graphlily_amd import capi
    from oracle import oracle as O
    exe = _build_typed_driver(define)
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "TYPED DRIVER DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    vt = capi.GL_VAL_UFIXED_32_8 if define == "UFIXED" else capi.GL_VAL_UNSIGNED

    def rd(name, dtype=np.uint32):
        return np.fromfile(str(tmp_path / (name + ".u32")), dtype=dtype)

    indptr, indices = rd("csr_indptr"), rd("csr_indices")
    data = O.words_from_float(vt, rd("csr_data_float", np.float32))
    x, mask, zero = rd("x"), rd("mask"), rd("zero_words")
    assert np.array_equal(x, O.words_from_float(vt, rd("x_float", np.float32))), "host float -> val_t conversion"
    assert list(zero) == [0, 0, 255 << 24 if define == "UFIXED" else 255]
    n = len(indptr) - 1
    saturated = 0
    for s in range(3):
        for k in range(3):
            ref = O.spmv_words(indptr, indices, data, x, s, vt, int(zero[s]), mask if k else None, k)
            got = rd("spmv_%d_%d" % (s, k))
            assert np.array_equal(got, ref), "SpMV op %d mask %d: %d rows differ" % (s, k, int((got != ref).sum()))
            assert len(np.unique(ref)) > 2 or s == 1
            saturated += int((ref == 0xffffffff).sum()) if s == 0 else 0
    if define == "UFIXED":
        assert saturated > 0, "the hub row's (+,x) sum saturates"
    cptr, cidx = rd("csc_indptr"), rd("csc_indices")
    cdata = O.words_from_float(vt, rd("csc_data_float", np.float32))
    sv = rd("sv").view(O.IDX_WORD)
    for s in range(3):
        for k in range(3):
            ref = O.spmspv_words(cptr, cidx, cdata, sv, n, s, vt, int(zero[s]), mask if k else None, k)
            res = rd("spmspv_%d_%d" % (s, k)).view(O.IDX_WORD)
            cnt = int(res["index"][0])
            got = np.full(n, zero[s], np.uint32)
            got[res["index"][1:cnt + 1]] = res["val"][1:cnt + 1]
            assert len(np.unique(res["index"][1:cnt + 1])) == cnt
            assert np.array_equal(got, ref), "SpMSpV op %d mask %d: %d rows differ" % (s, k, int((got != ref).sum()))
            assert cnt == int((ref != zero[s]).sum())
    # run_assign: (||,&&) WriteToZero, then the result rows of the mask vector set to 9
    ref = O.spmspv_words(cptr, cidx, cdata, sv, n, 1, vt, 0, mask, 1)
    want = mask.copy()
    want[ref != 0] = O.words_from_float(vt, [9.0])[0]
    assert np.array_equal(rd("run_assign_inout"), want)
    # apply modules
    assert np.array_equal(rd("ewise_out"), O.ewise_add_words(vt, rd("ewise_in"), int(rd("ewise_val")[0])))
    for k in (1, 2):
        want = rd("dense_inout_before").copy()
        O.assign_dense_words(k, rd("dense_mask"), want, int(rd("dense_val")[0]))
        assert np.array_equal(rd("dense_inout_after_%d" % k), want)
    ms = rd("sparse_mask").view(O.IDX_WORD)
    want = rd("sparse_io1_before").copy()
    want[ms["index"][1:int(ms["index"][0]) + 1]] = rd("sparse_val")[0]
    assert np.array_equal(rd("sparse_io1_after"), want)
    want = rd("sparse_io2_before").copy()
    nf = O.assign_sparse_new_frontier_words(ms, want)
    assert np.array_equal(rd("sparse_io2_after"), want)
    got_nf = rd("sparse_new_frontier").view(O.IDX_WORD)
    c = int(got_nf["index"][0])
    assert c == int(nf["index"][0]) and c > 0
    order = np.argsort(got_nf["index"][1:c + 1], kind="stable")
    assert np.array_equal(got_nf["index"][1:c + 1][order], nf["index"][1:c + 1])
    assert np.array_equal(got_nf["val"][1:c + 1][order], nf["val"][1:c + 1])


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_APPS), reason="prebuilt reference drivers did not travel")
def test_reference_app_drivers_run_on_hip_backend(gpu, tmp_path, golden_dir):
    import scipy.sparse as sp
    from graphlily_amd import datasets
    m = datasets.uniform(10000, 10, seed=7)
    A = sp.csr_matrix((m.adj_data, m.adj_indices.astype(np.int32), m.adj_indptr.astype(np.int32)),
                      shape=(m.num_rows, m.num_cols), dtype=np.float32)
    p = str(tmp_path / "uniform_10K_10_csr_float32.npz")
    sp.save_npz(p, A)
    for exe in (REF_APPS, REF_APPS + "_ufixed"):       # val_t = float, and the reference's shipped ap_ufixed<32, 8>
        for path in (p, os.path.join(golden_dir, "line_8_csr_float32.npz")):
            r = subprocess.run([exe, path], capture_output=True, text=True, timeout=600)
            print(r.stdout[-2000:])
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            assert r.stdout.count(" OK") == 7


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "bench_bfs_on_hip")),
                    reason="prebuilt reference benchmark drivers did not travel")
def test_reference_benchmark_drivers_run_on_hip_backend(gpu):
    """benchmark/bench_{spmv,bfs,pagerank,sssp}.cpp of the reference, compiled unmodified, run end to end; bench_spmv.cpp's
    own verify (:15-33, eps 1e-4 absolute) is called on the backend's results by tests/cpp/ref_bench_spmv_verify.cpp."""
    r = subprocess.run(["python", os.path.join(ROOT, "benchmarks", "run_reference_benches.py"),
                        "--graph", "googleplus", "--scale", "0.25"], capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("GTEPS") >= 6      # SpMV, BFS pull + pull-push, PageRank, SSSP pull + pull-push
    assert "SpMV passed" in r.stdout and "Compute THROUGHPUT" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_test_app")),
                    reason="prebuilt reference test suites did not travel")
def test_reference_test_suites_pass_on_hip_backend(gpu):
    """tests/test_module_apply.cpp:54-261, tests/test_module_spmv_spmspv.cpp:137-314 and tests/test_app.cpp:51-135 -- the files
    that DEFINE the acceptance bar (eps 1e-4, the semiring x mask matrix, the three apps on uniform_10K_10) -- byte-identical to
    the checkout, on the HIP backend, for val_t = float and for the reference's shipped ap_ufixed<32, 8>: every TEST passes
    (Synthesize is a no-op print, Clean removes ./proj in a scratch directory)."""
    r = subprocess.run(["python", os.path.join(ROOT, "benchmarks", "run_reference_benches.py"), "--apps", "tests",
                        "--write-reference-dataset-dir"], capture_output=True, text=True, timeout=1800)
    sys.stdout.write(r.stdout[-6000:] + r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "# 6 of 6 suite binaries passed" in r.stdout and "[  FAILED  ]" not in r.stdout
    # every TEST of every suite ran: 7 + 4 + 5 per value type (incl. Synthesize and Clean)
    assert r.stdout.count("[       OK ]") == 2 * (7 + 4 + 5)


@pytest.mark.gpu
def test_cpp_spmspv_sweep_driver(gpu, tmp_path):
    """benchmarks/bench_spmspv_cpp.cpp: the reference's bench_spmspv.cpp protocol through the C++ module layer (blocking run()
    on the operator's completion record); every case is verified against SpMSpVModule::compute_reference_results."""
    import json
    import scipy.sparse as sp
    from graphlily_amd import datasets
    sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
    import run_spmspv_cpp
    exe = run_spmspv_cpp.build()
    m = datasets.rmat(30000, 400000, seed=3)
    A = sp.csr_matrix((m.adj_data, m.adj_indices.astype(np.int32), m.adj_indptr.astype(np.int32)), shape=(m.num_rows, m.num_cols),
                      dtype=np.float32)
    p = str(tmp_path / "rmat_csr_float32.npz")
    sp.save_npz(p, A, compressed=False)
    r = subprocess.run([exe, "rmat", p, "0.9", "0.99", "0.999", "0.9999"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(rows) == 8 and all(x["verified"] for x in rows)
    assert {x["semiring"] for x in rows} == {"Arithmetic", "Tropical"}
