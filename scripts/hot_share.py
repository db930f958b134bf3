"""scratch: share of non-zeros held by the K highest-degree columns of a stand-in, and by columns of degree >= d"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import datasets
for g in sys.argv[1:]:
    if g.startswith("community"):
        m = datasets.community_torch(shuffle=g.endswith("shuffled"), device=torch.device("cuda:0"))
    else:
        m = datasets.paper_graph(g, 1.0, device=torch.device("cuda:0"))
    deg = np.bincount(m.adj_indices, minlength=m.num_cols)
    s = np.sort(deg)[::-1]
    c = np.cumsum(s) / m.nnz
    print(g, "n=%d nnz=%d maxdeg=%d" % (m.num_rows, m.nnz, s[0]),
          " ".join("top%dK=%.3f(d>=%d)" % (k // 1024, c[k - 1], s[k - 1]) for k in (1024, 4096, 8192, 16384, 24576, 32768, 49152, 65536, 131072, 262144) if k <= len(s)), flush=True)
