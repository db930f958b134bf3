"""Offline (CPU, numpy) check of the row-clustering idea of round 2's verdict: would letting a row block hold NON-contiguous rows
-- rows chosen so that their cold columns share 128-byte lines of the packed gather vector -- cut the lines a block's sweep
touches?  Measured on the R-MAT stand-in at reduced scale with the plan's own ingredients: hot columns (the top-degree columns
that fit the LDS table, scaled), the packed vector in degree classes, 256 nnz-balanced row blocks.  Three row orders:
  contiguous      what the planner does (rows in label order -- random labels);
  by min column   rows sorted by the smallest packed index among their cold columns (a cheap locality key);
  by minhash      rows sorted by a 4-way minhash of the LINES their cold columns fall into (neighbours share lines w.h.p.).
Prints distinct lines touched per block sweep (mean over blocks) and the ratio to contiguous.  usage: [scale] [graph]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import datasets

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
name = sys.argv[2] if len(sys.argv) > 2 else "orkut"
m = datasets.paper_graph(name, scale)      # (round 4: also the stand-ins with planted communities, datasets.EXTRA_GRAPHS)
n, nnz = m.num_rows, m.nnz
ip = m.adj_indptr.astype(np.int64)
cols = m.adj_indices.astype(np.int64)
row_of = np.repeat(np.arange(n), np.diff(ip))
deg = np.bincount(cols, minlength=n)
nblocks = 256
H = max(64, int(9216 * scale))                       # hot table: 9 K columns on the full orkut stand-in
hot = np.zeros(n, bool)
hot[np.argsort(-deg, kind="stable")[:H]] = True
# packed order: never-gathered columns dropped, the rest by degree class (>= nb/4, /16, /64, below), ascending inside
cls = np.full(n, -1)
gath = (deg > 0) & ~hot
edges = [max(nblocks // 4, 1), max(nblocks // 16, 1), max(nblocks // 64, 1)]
cls[gath] = np.where(deg[gath] >= edges[0], 0, np.where(deg[gath] >= edges[1], 1, np.where(deg[gath] >= edges[2], 2, 3)))
order = np.lexsort((np.arange(n), cls))
order = order[cls[order] >= 0]
packed = np.full(n, -1)
packed[order] = np.arange(order.shape[0])
cold = ~hot[cols]
line = packed[cols] // 32                            # 128-byte line of the packed vector an entry gathers from
print("%s x%.2f: n %d nnz %d, hot columns %d serve %.1f %% of the entries, packed vector %d columns = %d lines" % (
    name, scale, n, nnz, H, 100.0 * (~cold).mean(), order.shape[0], (order.shape[0] + 31) // 32))


def lines_per_block(row_rank):
    """row_rank[r] = position of row r in the order the blocks are cut from (nnz-balanced cuts)"""
    rows_in_order = np.argsort(row_rank, kind="stable")
    w = np.diff(ip)[rows_in_order]
    cum = np.cumsum(w)
    block_of_pos = np.minimum((cum - 1) * nblocks // max(nnz, 1), nblocks - 1)
    block_of_row = np.empty(n, np.int64)
    block_of_row[rows_in_order] = block_of_pos
    b = block_of_row[row_of][cold]
    key = b * (line.max() + 2) + line[cold]
    uniq = np.unique(key)
    per_block = np.bincount(uniq // (line.max() + 2), minlength=nblocks)
    return per_block


base = lines_per_block(np.arange(n))
# key 1: smallest packed index among the row's cold columns
big = np.iinfo(np.int64).max
mn = np.full(n, big)
np.minimum.at(mn, row_of[cold], packed[cols][cold])
k1 = lines_per_block(np.argsort(np.argsort(mn, kind="stable"), kind="stable"))
# key 2: 4-way minhash over the lines of the row's cold columns
rng = np.random.default_rng(0)
sig = []
for _ in range(4):
    a, c = int(rng.integers(1, 1 << 31)) | 1, int(rng.integers(0, 1 << 31))
    h = (line[cold] * a + c) % 2147483647
    s = np.full(n, big)
    np.minimum.at(s, row_of[cold], h)
    sig.append(s)
k2 = lines_per_block(np.argsort(np.argsort(np.lexsort(sig[::-1]), kind="stable"), kind="stable"))
total_lines = (order.shape[0] + 31) // 32
for label, v in (("contiguous (planner)", base), ("rows by min packed column", k1), ("rows by 4-way minhash of lines", k2)):
    print("%-32s lines touched per block sweep: mean %8.0f  max %8.0f  (%.1f %% of the packed vector's lines; x%.3f of contiguous)" % (
        label, v.mean(), v.max(), 100.0 * v.mean() / total_lines, v.mean() / base.mean()))
