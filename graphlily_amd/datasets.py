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


# A stand-in WITH structure (round 4): the R-MAT stand-ins above are randomly relabelled, so a row's columns are spread over the
# whole vector by construction -- every layout decision taken on them (no row clustering, the packed gather vector) rests on
# graphs without locality.  Real social / co-purchase graphs have communities; this one plants them.
EXTRA_GRAPHS = {
    "orkut_community":    dict(n=3_072_441, nnz=213_000_000, seed=16, symmetric=True, iters=6, kind="community"),
    "products_community": dict(n=2_449_029, nnz=124_000_000, seed=15, symmetric=True, iters=23, kind="community"),
    # the SAME graph with its vertices relabelled at random: what the community numbering is worth to a layout
    "orkut_community_shuffled": dict(n=3_072_441, nnz=213_000_000, seed=16, symmetric=True, iters=6, kind="community", shuffle=True),
    # no degree skew at all (round 5's verdict: the layouts' two big levers -- hot table, degree-class packed vector -- are tuned on
    # R-MAT skew): every row draws 70 columns uniformly.  The worst case for both levers at orkut's size.
    "uniform_3M_70": dict(n=3_072_441, nnz=215_000_000, seed=17, symmetric=False, iters=6, kind="uniform"),
}


def uniform_torch(n, nnz, seed, device):
    """n x n, nnz / n uniformly drawn columns per row (duplicates removed), built on the GPU like rmat_torch; host CSRMatrix."""
    import torch
    g = torch.Generator(device=device if device is not None else "cpu")
    g.manual_seed(seed)
    deg = max(1, nnz // n)
    dev = device if device is not None else torch.device("cpu")
    rows = torch.arange(n, device=dev, dtype=torch.int64).repeat_interleave(deg)
    cols = torch.randint(0, n, (n * deg,), generator=g, device=dev, dtype=torch.int64)
    key = torch.unique(rows * n + cols)
    del rows, cols
    rows = torch.div(key, n, rounding_mode="floor")
    cols = (key - rows * n).to(torch.int32)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    h_cols = cols.cpu().numpy().view(np.uint32)
    return CSRMatrix(n, n, np.ones(h_cols.shape[0], dtype=np.float32), h_cols, indptr.cpu().numpy().astype(np.uint32))


def community_torch(n, nnz, seed, symmetric, device, p_in=0.8, mean_size=2048, shuffle=False):
    """Degree-corrected stochastic block model: vertex weights from a power law (Pareto, exponent 2.2, capped), communities of
    geometrically spread sizes around `mean_size` numbered CONTIGUOUSLY (vertex ids follow the communities, as crawl order or
    a clustering pass leaves them), every edge picks its first endpoint by weight and its second one, with probability p_in,
    by weight inside the first one's community, else by weight anywhere.  Deduplicated; returns a host CSRMatrix."""
    import torch
    device = torch.device("cpu") if device is None else device
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    m = int((nnz // 2 if symmetric else nnz) * 1.12)
    # community boundaries: sizes mean_size * 2^U(-2, 2)
    sizes = []
    rs = np.random.default_rng(seed)
    total = 0
    while total < n:
        sz = max(16, int(mean_size * 2.0 ** rs.uniform(-2.0, 2.0)))
        sizes.append(min(sz, n - total))
        total += sizes[-1]
    bounds = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64, device=device)
    comm = torch.repeat_interleave(torch.arange(len(sizes), device=device), torch.tensor(sizes, device=device))
    u = torch.rand(n, generator=g, device=device, dtype=torch.float64)
    w = torch.clamp((1.0 - u) ** (-1.0 / 1.2), max=float(n) ** 0.5)        # Pareto tail, hubs capped at sqrt(n) x the minimum
    cum = torch.cumsum(w, 0)
    cum0 = torch.cat([torch.zeros(1, dtype=cum.dtype, device=device), cum])       # cum0[v] = weight in front of vertex v
    src = torch.searchsorted(cum, torch.rand(m, generator=g, device=device, dtype=torch.float64) * cum[-1]).clamp_(max=n - 1)
    inside = torch.rand(m, generator=g, device=device) < p_in
    c = comm[src]
    lo = torch.where(inside, cum0[bounds[c]], torch.zeros_like(cum0[bounds[c]]))
    hi = torch.where(inside, cum0[bounds[c + 1]], cum[-1].expand_as(lo))
    r = lo + torch.rand(m, generator=g, device=device, dtype=torch.float64) * (hi - lo)
    dst = torch.searchsorted(cum, r).clamp_(max=n - 1)
    del inside, c, lo, hi, r, u
    keep = src != dst
    src, dst = src[keep], dst[keep]
    if shuffle:
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
    h_cols = cols.cpu().numpy().view(np.uint32)
    return CSRMatrix(n, n, np.ones(h_cols.shape[0], dtype=np.float32), h_cols, indptr.cpu().numpy().astype(np.uint32))


def paper_graph(name, scale=1.0, device=None):
    """Stand-in for one of the six benchmark graphs (or one of EXTRA_GRAPHS); `scale` < 1 shrinks vertices and edges alike."""
    g = PAPER_GRAPHS.get(name) or EXTRA_GRAPHS[name]
    n = max(128, int(g["n"] * scale))
    nnz = max(1024, int(g["nnz"] * scale))
    if g.get("kind") == "community":
        return community_torch(n, nnz, g["seed"], g["symmetric"], device, shuffle=bool(g.get("shuffle")))
    if g.get("kind") == "uniform":
        return uniform_torch(n, nnz, g["seed"], device)
    if device is not None:
        return rmat_torch(n, nnz, g["seed"], g["symmetric"], device)
    return rmat(n, nnz, g["seed"], g["symmetric"])
