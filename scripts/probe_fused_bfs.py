import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from graphlily_amd import capi, datasets, io
dev = torch.device("cuda:0"); capi.init(0); capi.set_stream(torch.cuda.current_stream().cuda_stream)
m = datasets.paper_graph("orkut", 1.0, device=dev); io.util_round_csr_matrix_dim(m, 128, 128)
n = m.num_rows
plan = capi.SpMVPlan(n, n, m.adj_indptr, m.adj_indices, np.ones(m.nnz, np.float32), flags=capi.GL_PLAN_BOOLEAN)
words = plan.bits_words()
for dens_x, dens_d in ((0.01, 0.0), (0.3, 0.3), (0.05, 0.9)):
    x = (torch.rand(n, device=dev) < dens_x).float(); dist = (torch.rand(n, device=dev) < dens_d).float() * 2
    y = torch.zeros(n, device=dev); b1 = torch.zeros(words, device=dev); b2 = torch.zeros(words, device=dev)
    bx, bd, by, bb1, bb2 = (capi.DeviceBuffer.from_torch(t) for t in (x, dist, y, b1, b2))
    capi.pack_bits(bx, n, bb1)
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 30
    d0 = dist.clone()
    def fused():
        dist.copy_(d0); plan.bfs_pull_step(bb1, bb2, bd, 5.0)
    def copy_only(): dist.copy_(d0)
    print("x %.2f visited %.2f: run_bits masked %.3f ms | run (with pack) %.3f | fused step %.3f (minus dist reset %.3f)" % (
        dens_x, dens_d, t(lambda: plan.run_bits(bb1, bd, by, 0.0, 1)), t(lambda: plan.run(bx, bd, by, 1, 0.0, 1)), t(fused) - t(copy_only), t(copy_only)))
