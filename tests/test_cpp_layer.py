"""The C++ side of the drop-in boundary: include/graphlily/ (header-only module layer over the C ABI).

CPU: the parity driver compiles with plain g++ -std=c++11, links the HIP library and fails loudly without
a GPU; where /root/reference exists, the reference's own app drivers compile UNMODIFIED against these
headers (oracle/Makefile `ref_apps`).
GPU: the parity driver runs every module against its compute_reference_results; the prebuilt reference
drivers (if they travelled with the snapshot) run BFS / PageRank / SSSP end to end on the HIP backend."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "graphlily_amd", "lib")
DRIVER = os.path.join(ROOT, "build", "modules_driver")
REF_APPS = os.path.join(ROOT, "oracle", "_ref", "ref_apps_on_hip")


def _build_driver():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "modules_driver.cpp"), "-o", DRIVER,
                           "-L", LIBDIR, "-lgraphlily_hip", "-Wl,-rpath," + LIBDIR])


def test_module_headers_compile_and_fail_loudly_without_gpu():
    from graphlily_amd import capi
    _build_driver()
    if capi.device_count() == 0:
        r = subprocess.run([DRIVER], capture_output=True, text=True)
        assert r.returncode != 0
        assert "gl_init" in r.stdout + r.stderr       # print-and-exit convention of the reference


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
@pytest.mark.skipif(not os.path.exists(REF_APPS), reason="prebuilt reference drivers did not travel")
def test_reference_app_drivers_run_on_hip_backend(gpu, tmp_path, golden_dir):
    import scipy.sparse as sp
    from graphlily_amd import datasets
    m = datasets.uniform(10000, 10, seed=7)
    A = sp.csr_matrix((m.adj_data, m.adj_indices.astype(np.int32), m.adj_indptr.astype(np.int32)),
                      shape=(m.num_rows, m.num_cols), dtype=np.float32)
    p = str(tmp_path / "uniform_10K_10_csr_float32.npz")
    sp.save_npz(p, A)
    for path in (p, os.path.join(golden_dir, "line_8_csr_float32.npz")):
        r = subprocess.run([REF_APPS, path], capture_output=True, text=True, timeout=600)
        print(r.stdout[-2000:])
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert r.stdout.count(" OK") == 7


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "bench_bfs_on_hip")),
                    reason="prebuilt reference benchmark drivers did not travel")
def test_reference_benchmark_drivers_run_on_hip_backend(gpu):
    """benchmark/bench_{bfs,pagerank,sssp}.cpp of the reference, compiled unmodified, run end to end."""
    r = subprocess.run(["python", os.path.join(ROOT, "benchmarks", "run_reference_benches.py"),
                        "--graph", "googleplus", "--scale", "0.25"], capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("GTEPS") >= 5      # BFS pull + pull-push, PageRank, SSSP pull + pull-push
