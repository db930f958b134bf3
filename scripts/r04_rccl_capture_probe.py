"""One-GPU probe: does a grouped ncclSend / ncclRecv (to this same rank) complete when enqueued on the library's stream, and
when recorded by a stream capture and replayed?  (GRAPHLILY_DEBUG=dist_self_probe=1; run under `timeout`.)"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GRAPHLILY_DEBUG", "dist_self_probe=1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
from graphlily_amd import capi  # noqa: E402
from graphlily_amd.dist import CabiComm  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
capi.init(0)
capi.reset_stream()
comm = CabiComm(True, force=True)
n = 1 << 16
bits = capi.DeviceBuffer(n // 8)
bits.write(np.arange(n // 32, dtype=np.uint32))
bounds = [0, n]
t = time.time()
comm.exchange_bits(bits, 0, bounds)
capi.sync()
print("enqueued on the stream: done in %.3f s" % (time.time() - t), flush=True)
t = time.time()
with capi.Graph.capture() as g:
    comm.exchange_bits(bits, 0, bounds)
print("recorded: %.3f s" % (time.time() - t), flush=True)
t = time.time()
for _ in range(3):
    g.launch()
capi.sync()
print("replayed 3x: done in %.3f s" % (time.time() - t), flush=True)
