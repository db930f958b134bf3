"""Deterministic synthetic stand-ins for the matrices the reference's tests and benchmarks name.

The reference keeps its matrices on the authors' NFS (tests/test_module_spmv_spmspv.cpp:144-145,
167-168, 251-266) and its benchmark graphs behind download links (README.md:44-49); none are
available offline.  These generators recreate them by name and size:

  dense(n)                    dense_32 / dense_1K: n x n, all ones
  uniform(n, deg, seed)       uniform_10K_10: deg distinct uniformly random columns per row
  conflict(n)                 the bank-conflict CSC of tests/test_module_spmv_spmspv.cpp:268-284
  rmat(...)                   power-law graphs with the paper graphs' vertex / edge counts
  PAPER_GRAPHS                name -> (vertices, edges, seed, symmetric, bfs/sssp iterations)

Everything is returned as graphlily_amd.io.CSRMatrix (or CSCMatrix for conflict()).
"""
import numpy as np

from .io import CSCMatrix, CSRMatrix

# benchmark/run_spmv.sh:12-17 (sizes), benchmark/run_bfs.sh:20 (iterations); seeds from SURVEY.md 8d
PAPER_GRAPHS = {
    "googleplus":    dict(n=108_000,   nnz=13_000_000,  seed=1, symmetric=False, iters=7),
    "ogbl_ppa":      dict(n=576_000,   nnz=42_000_000,  seed=2, symmetric=True,  iters=11),
    "hollywood":     dict(n=1_000_000, nnz=113_000_000, seed=3, symmetric=True,  iters=10),
    "pokec":         dict(n=1_633_000, nnz=31_000_000,  seed=4, symmetric=False, iters=11),
    "ogbn_products": dict(n=2_449_029, nnz=124_000_000, seed=5, symmetric=True,  iters=23),
    "orkut":         dict(n=3_072_441, nnz=213_000_000, seed=6, symmetric=True,  iters=6),
}


def _csr_from_sorted_pairs(n_rows, n_cols, rows, cols, vals=None):
    counts = np.bincount(rows, minlength=n_rows)
    indptr = np.zeros(n_rows + 1, dtype=np.uint32)
    np.cumsum(counts, out=indptr[1:], dtype=np.uint64)
    if vals is None:
        vals = np.ones(rows.shape[0], dtype=np.float32)
    return CSRMatrix(n_rows, n_cols, vals, cols.astype(np.uint32), indptr)


def dense(n):
    cols = np.tile(np.arange(n, dtype=np.uint32), n)
    indptr = np.arange(0, n * n + 1, n, dtype=np.uint32)
    return CSRMatrix(n, n, np.ones(n * n, dtype=np.float32), cols, indptr)


def uniform(n, deg, seed=7):
    rng = np.random.default_rng(seed)
    cols = np.empty((n, deg), dtype=np.int64)
    for r in range(n):
        cols[r] = np.sort(rng.choice(n, size=deg, replace=False))
    indptr = np.arange(0, n * deg + 1, deg, dtype=np.uint32)
    return CSRMatrix(n, n, np.ones(n * deg, dtype=np.float32), cols.reshape(-1).astype(np.uint32), indptr)


def conflict(n=1024):
    """Column i holds rows j*8 + i%8, j in [0, n/8): every column hammers one residue class, the
    worst case for the FPGA's banked output buffer and, here, for same-address atomics."""
    per_col = n // 8
    i = np.repeat(np.arange(n, dtype=np.uint32), per_col)
    j = np.tile(np.arange(per_col, dtype=np.uint32), n)
    rows = j * 8 + (i % 8)
    indptr = np.arange(0, n * per_col + 1, per_col, dtype=np.uint32)
    return CSCMatrix(n, n, np.full(n * per_col, 1.0 / n, dtype=np.float32), rows, indptr)


def rmat_edges_numpy(n, m, seed, a=0.57, b=0.19, c=0.19):
    """m directed R-MAT edges over n vertices (ids scaled from the enclosing power of two and
    relabelled by a random permutation so that hubs are not clustered at low ids)."""
    rng = np.random.default_rng(seed)
    k = max(1, int(np.ceil(np.log2(n))))
    src = np.zeros(m, dtype=np.int64)
    dst = np.zeros(m, dtype=np.int64)
    for _ in range(k):
        u = rng.random(m)
        src = (src << 1) | (u >= a + b)
        dst = (dst << 1) | (((u >= a) & (u < a + b)) | (u >= a + b + c))
    src = (src * n) >> k
    dst = (dst * n) >> k
    perm = rng.permutation(n)
    return perm[src], perm[dst]


def rmat(n, nnz, seed, symmetric=False):
    """Deduplicated R-MAT graph as CSR with unit values; nnz is a target (dedupe / symmetrisation
    change it by a few percent)."""
    m = nnz // 2 if symmetric else nnz
    src, dst = rmat_edges_numpy(n, int(m * 1.08), seed)
    if symmetric:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    key = np.unique(src * n + dst)
    rows = (key // n).astype(np.int64)
    cols = (key % n).astype(np.int64)
    return _csr_from_sorted_pairs(n, n, rows, cols)


def rmat_torch(n, nnz, seed, symmetric, device):
    """Same construction on the GPU (torch is used as a sort/unique engine only) for the
    hundred-million-edge stand-ins; returns host CSRMatrix."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    m = int((nnz // 2 if symmetric else nnz) * 1.08)
    k = max(1, int(np.ceil(np.log2(n))))
    a, b, c = 0.57, 0.19, 0.19
    src = torch.zeros(m, dtype=torch.int64, device=device)
    dst = torch.zeros(m, dtype=torch.int64, device=device)
    for _ in range(k):
        u = torch.rand(m, generator=g, device=device)
        src = (src << 1) | (u >= a + b).long()
        dst = (dst << 1) | (((u >= a) & (u < a + b)) | (u >= a + b + c)).long()
        del u
    src = (src * n) >> k
    dst = (dst * n) >> k
    perm = torch.randperm(n, generator=g, device=device)
    src, dst = perm[src], perm[dst]
    if symmetric:
        src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    key = torch.unique(src * n + dst)
    del src, dst
    rows = torch.div(key, n, rounding_mode="floor")
    cols = (key - rows * n).to(torch.int32)
    counts = torch.bincount(rows, minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    h_cols = cols.cpu().numpy().view(np.uint32) if cols.dtype == torch.int32 else cols.cpu().numpy().astype(np.uint32)
    h_indptr = indptr.cpu().numpy().astype(np.uint32)
    return CSRMatrix(n, n, np.ones(h_cols.shape[0], dtype=np.float32), h_cols, h_indptr)


def paper_graph(name, scale=1.0, device=None):
    """Stand-in for one of the six benchmark graphs; `scale` < 1 shrinks vertices and edges alike."""
    g = PAPER_GRAPHS[name]
    n = max(128, int(g["n"] * scale))
    nnz = max(1024, int(g["nnz"] * scale))
    if device is not None:
        return rmat_torch(n, nnz, g["seed"], g["symmetric"], device)
    return rmat(n, nnz, g["seed"], g["symmetric"])
