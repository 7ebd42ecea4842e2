"""A 1-rank "nccl" (= RCCL) process group on the GPU box: loads RCCL, creates the sequence-parallel
group the way the pipeline does and runs every method of engine.TorchGroupComm once on device
tensors, plus `generate(use_usp=True, sp_size=1)`.  One GPU cannot host two RCCL ranks, so this is the
most the one-GPU box can execute of the RCCL path; the multi-rank logic is covered over gloo."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "omni-video_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import _sp_case as case                              # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="env://")
    from omnivideo.distributed import util as U
    from omnivideo.modules.engine import TorchGroupComm, make_group_comm
    group = U.init_sequence_parallel_group(1)
    comm = make_group_comm(group)
    assert isinstance(comm, TorchGroupComm) and (comm.size, comm.rank) == (1, 0)
    g = torch.Generator(device="cuda").manual_seed(0)
    send = torch.randn(1, 3, 2, 5, 128, device="cuda", generator=g).to(torch.bfloat16)
    recv = torch.zeros_like(send)
    comm.all_to_all(recv, send)
    # split-phase form (what the overlapped sequence-parallel forward issues): queued, other work, then waited for
    recv2, recv3 = torch.zeros_like(send), torch.zeros_like(send)
    comm.all_to_all_start(recv2, send, ("qkv", 0, 0))
    comm.all_to_all_start(recv3, send.flip(1).contiguous(), ("qkv", 1, 0))
    busy = (send.float() @ send.float().transpose(-1, -2)).sum()
    comm.wait(("qkv", 0, 0))
    comm.wait(("qkv", 1, 0))
    torch.cuda.synchronize()
    assert torch.equal(recv2, send) and torch.equal(recv3, send.flip(1)) and torch.isfinite(busy)
    t = torch.randn(2, 16, 3, 8, 12, device="cuda", generator=g)
    t0 = t.clone()
    comm.all_reduce_sum(t)
    up, down = comm.exchange_rows(t[0, 0, 0], t[0, 0, 1])
    rows = comm.all_gather_rows(t[0], [t.shape[2]])
    y = U.all_to_all(send[0], scatter_dim=2, gather_dim=1, group=group)
    gathered = U.gather_forward(send, dim=1, group=group)
    torch.cuda.synchronize()
    assert torch.equal(recv, send) and torch.equal(t, t0) and up is None and down is None
    assert torch.equal(rows, t0[0]) and torch.equal(y, send[0]) and torch.equal(gathered, send)
    with torch.no_grad():
        pipe, cfg = case.build_pipeline(use_usp=True, sp_size=1)
        lat, video = case.run_generate(pipe, cfg)
        ref_pipe, _ = case.build_pipeline(use_usp=False)
        ref_lat, ref_video = case.run_generate(ref_pipe, cfg)
    assert torch.equal(lat, ref_lat) and torch.equal(video, ref_video)
    dist.barrier()
    dist.destroy_process_group()
    print("rank 0 ok")


if __name__ == "__main__":
    main()
