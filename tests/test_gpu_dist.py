"""GPU suite: the N > 1 path with the REAL device modules.  Two ranks over gloo share cuda:0 (the GPU box has one
device; RCCL needs one device per rank), so the collectives are host-staged by Comm, everything else -- row-shard
plans, bit-vector frontier exchange of the BFS pull, sparse frontier all-gather of the push, on-device direction
switch, split-plan combine -- is the code an 8-GPU run executes.  Results must equal the oracle's on every rank."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from graphlily_amd import datasets
from graphlily_amd.dist import Comm
from oracle import oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph():
    return datasets.rmat(40000, 900000, seed=23, symmetric=True)


def _run_apps(comm, which):
    from graphlily_amd import app, module as M
    m = _graph()
    B = app.HipBackend(0, use_torch=True)
    if which == "bfs":
        a = app.BFS(16, 0, 0, 0, comm=comm, backend=B)
        a.set_up_runtime()
        a.load_and_format_matrix(m, True)
        a.send_matrix_host_to_device()
        out = [a.pull(0, 7), a.pull_push(0, 7, 0.001), a.pull_push(0, 7, 0.5)]
        assert a.bits_loop_ is not None, "a sharded BFS runs the device-resident bit-frontier schedule (bit vectors exchanged)"
        return out
    if which == "pagerank":
        a = app.PageRank(16, 0, 0, comm=comm, backend=B)
        a.set_up_runtime()
        a.load_and_format_matrix(m, 0.9, True)
        a.send_matrix_host_to_device()
        return [a.pull(0.9, 6)]
    a = app.SSSP(16, 0, 0, 0, comm=comm, backend=B, semiring=M.TropicalSemiringUfixed)
    a.set_up_runtime()
    a.load_and_format_matrix(m, True)
    a.send_matrix_host_to_device()
    return [a.pull(0, 7), a.pull_push(0, 7, 0.01)]


def _worker(rank, world, port, which, out_q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out_q.put((rank, _run_apps(Comm(True), which)))
    finally:
        dist.destroy_process_group()


def _spawn(which, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, which, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    deadline = time.time() + 300
    while len(out) < world and time.time() < deadline:   # drain before joining (a put() blocks until read)
        if not q.empty():
            r, res = q.get()
            out[r] = res
        elif any(p.exitcode not in (None, 0) for p in procs):
            break
        else:
            time.sleep(0.05)
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0, "worker failed (exit code %s)" % p.exitcode
    assert len(out) == world
    return out


def _oracle_graph(sssp=False, pagerank=False):
    m = _graph()
    om = O.CSR(m.num_rows, m.num_cols, m.adj_data, m.adj_indices, m.adj_indptr)
    if sssp:
        O.sssp_preprocess(om)
    O.util_round_csr_matrix_dim(om, 128, 128)
    if pagerank:
        O.util_normalize_csr_matrix_by_outdegree(om)
        om.adj_data = (om.adj_data * np.float32(0.9)).astype(np.float32)
    elif not sssp:
        om.adj_data[:] = 1
    return om


def test_two_ranks_bfs(gpu):
    ref = O.bfs(_oracle_graph(), 0, 7)
    out = _spawn("bfs")
    for r in (0, 1):
        for got in out[r]:
            assert np.array_equal(got, ref)


def test_two_ranks_pagerank(gpu):
    ref = O.pagerank(_oracle_graph(pagerank=True), 0.9, 6)
    out = _spawn("pagerank")
    for r in (0, 1):
        np.testing.assert_allclose(out[r][0], ref, rtol=1e-4, atol=1e-9)
    assert np.array_equal(out[0][0], out[1][0])


def test_two_ranks_sssp(gpu):
    ref = O.sssp(_oracle_graph(sssp=True), 0, 7, 255.0)
    out = _spawn("sssp")
    for r in (0, 1):
        for got in out[r]:
            assert np.array_equal(got, ref)


def test_bench_self_launches_two_ranks(gpu):
    """`python bench.py --gpus 2` from a plain shell (no torch.distributed.run around it, WORLD_SIZE unset) starts its
    own ranks; here both share cuda:0 over gloo (the box has one GPU).  The JSON line must say n_gpus = 2."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-gpu",
                        "--steps", "20", "--warmup", "2", "--scale", "0.1", "--bfs-runs", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 20 and rec["selfcheck_ok"] is True
    assert rec["scaling"] == "strong" and "roofline" in rec
    assert "error" not in rec.get("bfs", {}), rec["bfs"]


def test_bench_self_launches_eight_ranks(gpu):
    """The 8-rank launch the driver performs at round end, as far as one GPU goes: eight processes, eight row shards of the
    stand-in (scale 0.05), gloo with host-staged collectives, every rank on cuda:0.  The row-sharded SpMV step (uneven
    nnz-balanced ranges), the device-resident sharded BFS schedule with its per-slot all-gather (bits + tallies), and
    the max-over-ranks timing all run with world = 8."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--same-gpu",
                        "--steps", "5", "--warmup", "1", "--scale", "0.05", "--bfs-runs", "1", "--no-pattern"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["selfcheck_ok"] is True and rec["scaling"] == "strong"
    assert "error" not in rec.get("bfs", {}), rec["bfs"]
    assert rec["bfs"]["pull"]["reached"] == rec["bfs"]["pull_push"]["reached"] > 0


def test_bench_one_rank_rccl(gpu):
    """The RCCL leg of bench.py as far as one GPU allows: `--force-dist` creates the nccl process group for a world of
    one and runs every collective of the N > 1 path on it -- the in-place all_gather_into_tensor of the SpMV step, the
    max-over-ranks all_reduce, and the bit / sparse-list gathers of the row-sharded BFS driver."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "20", "--warmup", "2",
                        "--scale", "0.1", "--bfs-runs", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["selfcheck_ok"] is True
    assert "error" not in rec.get("bfs", {}), rec["bfs"]
    assert rec["bfs"]["pull"]["reached"] == rec["bfs"]["pull_push"]["reached"] > 0
    # the pre-flight ran on RCCL before anything was timed: ranks as asked, both verified all-gathers, the exchange timed apart
    assert rec["rccl_ranks"] == 1 and rec["exchange_path"]["backend"] == "nccl" and rec["exchange_path"]["verified"] is True
    assert rec["exchange_path"]["bfs"] == "torch" and rec["exchange_ms"]["torch"]["dense_12MB"] > 0
    assert rec["exchange_ms"]["per_app"]["bfs_per_run"] > 0


def test_bench_falls_back_when_the_c_abi_preflight_fails(gpu):
    """`--cabi-comm` with a pre-flight failure of the C ABI exchange (forced: GRAPHLILY_DEBUG dist_preflight_fail_cabi=1): the
    failure is recorded, the sharded BFS runs on the torch.distributed path, the line still comes out, and the process ends
    without waiting for a communicator (the C ABI communicator is destroyed right away)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["GRAPHLILY_DEBUG"] = "dist_preflight_fail_cabi=1"
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--cabi-comm", "--no-six-graphs", "--no-spmspv",
                        "--steps", "10", "--warmup", "2", "--scale", "0.1", "--bfs-runs", "1", "--no-cpu-baseline", "--no-pattern"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["exchange_path"]["bfs"] == "torch" and "forced by the test hook" in rec["exchange_path"]["cabi_error"]
    assert rec["bfs"]["exchange"] == "Comm" and "error" not in rec["bfs"]
    assert rec["bfs"]["pull"]["reached"] == rec["bfs"]["pull_push"]["reached"] > 0


def test_bench_one_rank_rccl_exchange_inside_the_graph(gpu):
    """`--cabi-comm`: the sharded BFS exchanges bits + tallies through the C ABI (gl_dist_all_gather_bits_tally: grouped
    ncclSend / ncclRecv on the library's stream), so the exchange is recorded INTO the schedule's hipGraph -- one launch and
    one exchange per slot, replayed with one call.  One GPU allows a world of one, whose exchange is empty; the planner
    override dist_self_probe=1 makes every exchange call put one grouped ncclSend / ncclRecv (to this same rank) on the stream,
    so that RCCL operations ARE recorded by the capture and replayed -- what a world of N relies on."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["GRAPHLILY_DEBUG"] = "dist_self_probe=1"
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--cabi-comm", "--no-six-graphs", "--no-spmspv", "--steps", "10", "--warmup", "2",
                        "--scale", "0.1", "--bfs-runs", "1", "--no-cpu-baseline", "--no-pattern"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert "error" not in rec.get("bfs", {}), rec["bfs"]
    assert rec["bfs"]["exchange"] == "CabiComm" and "hipGraph" in rec["bfs"]["schedule"], rec["bfs"]
    assert rec["bfs"]["pull"]["reached"] == rec["bfs"]["pull_push"]["reached"] > 0
    # the pre-flight verified the C ABI exchange (bits + tallies, dense) before the BFS leg took that path
    assert rec["exchange_path"]["bfs"] == "cabi" and rec["exchange_path"]["cabi_error"] is None
    assert rec["exchange_ms"]["cabi"]["bits_384KB_with_tallies"] > 0 and rec["exchange_ms"]["cabi"]["dense_12MB"] > 0


def test_a_communicator_outlives_the_graphs_that_recorded_it(gpu):
    """RCCL's communicator destroy waits for every graph that holds its operations: gl_dist_destroy refuses (instead of
    hanging) while a gl_graph that recorded one of the communicator's exchanges is alive."""
    import ctypes
    from graphlily_amd import capi
    d = capi.Dist(0, 1, capi.Dist.unique_id())
    n = 1 << 12
    bits = capi.DeviceBuffer.from_host(np.arange(n // 32, dtype=np.uint32))
    capi.reset_stream()
    with capi.Graph.capture() as g:
        d.all_gather_bits_tally(bits, [0, n], None)
    g.launch()
    capi.sync()
    rc = capi.lib().gl_dist_destroy(ctypes.c_void_p(d.handle))
    assert rc == capi.GL_ERR_INVALID_ARG and b"recorded graph" in capi.lib().gl_last_error()
    d.destroy()                          # (the wrapper ends the live graphs first)
    assert d.handle is None and g.handle is None
    with pytest.raises(capi.GraphLilyError):
        g.launch()


def test_gl_dist_c_abi_single_rank(gpu):
    """gl_dist_* (the RCCL exchange step in the C ABI): the box has one GPU, so this is a world of one -- RCCL loads at
    run time, ncclCommInitRank succeeds on the device, the dense / bit gathers leave the vector alone and the sparse gather
    builds the concatenated list with its head.  (The N > 1 exchange is the same grouped send / recv with more peers.)"""
    from graphlily_amd import capi
    d = capi.Dist(0, 1, capi.Dist.unique_id())
    n = 4096
    x = np.arange(n, dtype=np.float32)
    bx = capi.DeviceBuffer.from_host(x)
    d.all_gather_f32(bx, [0, n])
    bits = capi.DeviceBuffer.from_host(np.arange(n // 32, dtype=np.uint32))
    d.all_gather_bits_tally(bits, [0, n], None)
    capi.sync()
    assert np.array_equal(bx.read(np.float32, n), x)
    assert np.array_equal(bits.read(np.uint32, n // 32), np.arange(n // 32, dtype=np.uint32))
    local = np.zeros(n + 1, dtype=capi.IDX_VAL)
    local["index"][0] = 5
    local["index"][1:6] = [3, 9, 27, 81, 243]
    local["val"][1:6] = [1, 2, 3, 4, 5]
    bl, bf = capi.DeviceBuffer.from_host(local), capi.DeviceBuffer(8 * (n + 1))
    total = d.all_gather_sparse(bl, bf, n, 255.0)
    capi.sync()
    got = bf.read(capi.IDX_VAL, 6)
    assert total == 5 and got["index"][0] == 5 and got["val"][0] == 255.0
    assert np.array_equal(got[1:], local[1:6])
    with pytest.raises(capi.GraphLilyError):
        d.all_gather_bits_tally(bits, [0, 100, n], None) if False else capi.Dist(0, 1, b"short")   # a unique id is 128 bytes
    d.destroy()
