#!/usr/bin/env python
"""Barrier latency and exchange rate of the peer-mapped communicator between ranks that are processes on ONE
GPU (development box): python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
--master-port 29551 tools/peer_microbench.py.  The payload moves at HBM speed here; on a node the same
kernels read over xGMI."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omni-video_amd"))
from omnivideo.distributed.peer_mapped import PeerMappedComm      # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = PeerMappedComm(None, timeout_s=20)

    def timed(fn, iters):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    g = torch.cuda.CUDAGraph()
    comm.barrier()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(100):
            comm.barrier()
    us_graph = timed(g.replay, 10) * 10.0
    us_eager = timed(comm.barrier, 200) * 1e3
    lines = [f"barrier over {world} processes: {us_graph:.1f} us each inside a graph, {us_eager:.1f} us launched eagerly"]
    for mb in (1, 12, 96):            # bytes per (peer, chunk): 12 MB = one q / k / v slab per peer at 81f 720p over 8 ranks
        n = mb * (1 << 20) // 2
        send = comm.alloc((world, n), torch.bfloat16)
        recv = torch.empty(world, n, dtype=torch.bfloat16, device="cuda")
        ms = timed(lambda: comm.all_to_all(recv, send), 20)
        lines.append(f"all_to_all {mb} MB per peer: {ms * 1e3:.0f} us = {world * n * 2 / ms / 1e6:.0f} GB/s received per rank")
        comm.free(send)
    comm.check()
    if rank == 0:
        print("\n".join(lines), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
