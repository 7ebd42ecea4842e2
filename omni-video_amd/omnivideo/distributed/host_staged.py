"""Communicator for process groups whose backend moves host memory only (gloo): the interface of
`omnivideo.modules.engine.TorchGroupComm` with every device tensor staged through the host.

On MI355X nodes the sequence-parallel group runs on the "nccl" backend (RCCL over xGMI) and this
class is not used.  It exists so that the multi-rank control flow - row sharding, the two all-to-alls
per layer, the head all-reduce, the row-sharded VAE decode - can be executed by several processes
that share ONE GPU (RCCL refuses two ranks on one device): the 2-process GPU tests and
`bench.py --rehearse-on-one-gpu`.  It moves bytes; no arithmetic happens on the host except the
all-reduce sum, which adds exact zeros to one owner's values (every element has one owner rank)."""
import torch
import torch.distributed as dist


class HostStagedComm:
    def __init__(self, group=None):
        self.group = group
        self.size = dist.get_world_size(group=group)
        self.rank = dist.get_rank(group=group)

    def _peer(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def all_to_all(self, recv, send):
        src = send.cpu().contiguous().view(torch.uint8).reshape(self.size, -1)       # bytes: any element type
        r = torch.empty_like(src)
        dist.all_to_all_single(r, src, group=self.group)
        recv.copy_(r.view(recv.dtype).reshape(recv.shape))

    # split-phase form of engine.TorchGroupComm: nothing runs beside a host-staged exchange, so the
    # exchange happens at start and wait() has nothing left to do
    def all_to_all_start(self, recv, send, tag):
        self.all_to_all(recv, send)

    def wait(self, tag):
        pass

    def all_reduce_sum(self, t):
        c = t.cpu()
        dist.all_reduce(c, group=self.group)
        t.copy_(c)

    def exchange_rows(self, up, down):
        ops, from_up, from_down = [], None, None
        if self.rank > 0:
            from_up = torch.empty(up.shape, dtype=up.dtype)
            ops += [dist.P2POp(dist.isend, up.cpu(), self._peer(self.rank - 1), group=self.group),
                    dist.P2POp(dist.irecv, from_up, self._peer(self.rank - 1), group=self.group)]
        if self.rank < self.size - 1:
            from_down = torch.empty(down.shape, dtype=down.dtype)
            ops += [dist.P2POp(dist.isend, down.cpu(), self._peer(self.rank + 1), group=self.group),
                    dist.P2POp(dist.irecv, from_down, self._peer(self.rank + 1), group=self.group)]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return (None if from_up is None else from_up.to(up.device),
                None if from_down is None else from_down.to(down.device))

    def all_gather_rows(self, x, rows):
        mx = max(rows)
        pad = torch.zeros((x.shape[0], mx) + tuple(x.shape[2:]), dtype=x.dtype)
        pad[:, :x.shape[1]] = x.cpu()
        parts = [torch.empty_like(pad) for _ in range(self.size)]
        dist.all_gather(parts, pad, group=self.group)
        return torch.cat([p[:, :n] for p, n in zip(parts, rows)], dim=1).to(x.device)


def moves_host_memory_only(group=None):
    """True for a group on the gloo backend (its all-to-all / point-to-point paths take CPU tensors)."""
    return dist.is_initialized() and str(dist.get_backend(group)).lower() == "gloo"
