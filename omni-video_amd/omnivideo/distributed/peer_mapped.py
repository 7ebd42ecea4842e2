"""Sequence-parallel communicator over peer-mapped device memory (one process per GPU on one node).

The reference exchanges q, k, v and the attention output with `dist.all_to_all` on NCCL, wrapped in
chunk / contiguous / cat copies (omnivideo/distributed/util.py:95-121, ulysses.py:41-55), and gathers
the head output with `dist.all_gather` (util.py:124-130, full_sequence_parallel.py:163).  On an MI355X
node the eight GPUs form a full xGMI mesh, so an all-to-all is seven independent point-to-point
transfers per GPU, and that is how it is done here: every rank maps the exchange buffers of its peers
into its address space once (HIP IPC, `ovd_peer_*` set-up calls), a barrier kernel orders the ranks'
streams against each other, and one copy kernel reads this rank's chunk from all peers at once.  No
collective library is involved and nothing returns to the host, so the whole sampling step of a rank,
exchanges included, is captured as ONE HIP graph (`capturable`), where the RCCL communicator
(`engine.TorchGroupComm`) needs a chain of graphs with eager collectives between them.

Interface = `engine.TorchGroupComm`'s, plus
  alloc(shape, dtype)   COLLECTIVE: a tensor in exported memory.  A tensor passed as `send` to
                        `all_to_all` must come from here (peers read it in place);
  table(t, offset)      the addresses of such a tensor on every rank, for kernels that address the peers
                        themselves: with `direct` (default) the engine's RMSNorm+RoPE kernel writes q, k, v
                        straight into every peer's attention operands and its head-gather kernel reads the
                        attention outputs straight from the peers - two barriers per layer and no copy pass;
  barrier()             stream-ordered barrier over the group;
  check()               COLLECTIVE: raises on every rank if a barrier of any rank ever timed out (host
                        synchronisation; call it at the end of a run).
Ordering rule for callers: a `send` tensor may be rewritten once this rank has issued one more
collective of this communicator after the `all_to_all` that read it (the engine's layer loop does:
q/k/v are rewritten after the output exchange, the output after the next layer's q/k/v exchange).
The two eager exchanges of the row-sharded VAE decode are delegated to the group's ordinary
communicator.  Select with OVD_SP_COMM=peer; OVD_PEER_TIMEOUT_S bounds the wait of a barrier."""
import os
import socket

import torch
import torch.distributed as dist

from .. import _hip as H

F32 = torch.float32


class _Raw:
    """Exposes a raw device address range to torch (`torch.as_tensor` reads __cuda_array_interface__)."""

    def __init__(self, addr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (addr, False), "version": 2}


class _Region:
    def __init__(self, base, nbytes, peers, opened):
        self.base, self.nbytes, self.peers, self.opened = base, nbytes, peers, opened
        self.bytes = torch.as_tensor(_Raw(base, nbytes), device="cuda")


class PeerMappedComm:
    capturable = True          # every method issues stream work only: a step may be captured as one graph

    def __init__(self, group=None, timeout_s=None, base=None, direct=None, payload=None):
        """payload: "default" = exchange buffers in ordinary (coarse-grained) device memory: what a kernel wrote
        reaches the peers when the kernel ends (the release at the end of every kernel writes its dirty L2 lines
        back) and a reading kernel sees it because it starts after its own barrier kernel has ended (the acquire
        at the start of every kernel invalidates the non-coherent cache levels); "fine" = fine-grained memory,
        which the hardware keeps coherent across devices at every access (slower on the owning GPU: its own
        accesses bypass the L2 for such lines).  Default: OVD_PEER_PAYLOAD, else "default"."""
        from .host_staged import HostStagedComm, moves_host_memory_only
        payload = (payload or os.environ.get("OVD_PEER_PAYLOAD") or "default").lower()
        if payload not in ("default", "fine"):
            raise ValueError(f"OVD_PEER_PAYLOAD / payload must be 'default' or 'fine', got {payload!r}")
        self.payload = payload
        # direct: the engine's kernels write to / read from the peers themselves (two barriers per layer, no
        # copy pass); otherwise every exchange is barrier + all_to_all (OVD_PEER_DIRECT=0)
        self.direct = (os.environ.get("OVD_PEER_DIRECT", "1") != "0") if direct is None else bool(direct)
        self.group = group
        self.size = dist.get_world_size(group=group)
        self.rank = dist.get_rank(group=group)
        if self.size > 64:
            raise ValueError("the barrier kernel serves at most 64 ranks")
        if base is None:
            from ..modules.engine import TorchGroupComm
            base = HostStagedComm(group) if moves_host_memory_only(group) else TorchGroupComm(group)
        self.base = base
        self.timeout_s = float(os.environ.get("OVD_PEER_TIMEOUT_S", "30") if timeout_s is None else timeout_s)
        self._regions, self._tables = [], {}
        self._flags = self._new_region(512, fine_grained=True)   # 64 epoch words + 64 failure marks
        self._flag_table = self._table(self._flags, 0)
        self._state = torch.zeros(2, dtype=torch.int32, device="cuda")
        self._stage = None         # the newest (largest) staging region of all_reduce_sum; older ones stay mapped
        self.barriers = 0          # barrier launches issued (captured ones count once)

    # ------------------------------------------------------------------ exported memory
    def _new_region(self, nbytes, fine_grained=None):
        """COLLECTIVE.  Allocates `nbytes` of exportable device memory on every rank and maps every
        peer's buffer here."""
        nbytes = -(-int(nbytes) // 256) * 256
        if fine_grained is None:
            fine_grained = self.payload == "fine"
        base = H.peer_alloc(nbytes, fine_grained)
        mine = (os.getpid(), socket.gethostname(), base, nbytes, H.peer_export(base))
        infos = [None] * self.size
        dist.all_gather_object(infos, mine, group=self.group)
        peers, opened = [], []
        for j, (pid, host, addr, nb, handle) in enumerate(infos):
            if host != mine[1]:
                raise RuntimeError(f"rank {j} runs on {host}, this rank on {mine[1]}: peer mapping is for one node")
            if nb != nbytes:
                raise RuntimeError(f"rank {j} allocated {nb} bytes where this rank allocated {nbytes}: "
                                   "alloc() must be called with the same sizes in the same order on every rank")
            if j == self.rank:
                peers.append(base)
            elif pid == mine[0]:
                peers.append(addr)               # ranks as threads of one process share the address space
            else:
                a = H.peer_open(handle)
                peers.append(a)
                opened.append(a)
        reg = _Region(base, nbytes, peers, opened)
        self._regions.append(reg)
        return reg

    def _table(self, reg, offset):
        """Device array of the p addresses `offset` bytes into every rank's copy of a region."""
        key = (id(reg), int(offset))
        t = self._tables.get(key)
        if t is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("first use of an exchange buffer under graph capture: run the step once eagerly")
            t = torch.tensor([a + int(offset) for a in reg.peers], dtype=torch.int64, device="cuda")
            self._tables[key] = t
        return t

    def _region_of(self, t):
        a = t.data_ptr()
        for reg in self._regions:
            if reg.base <= a and a + t.numel() * t.element_size() <= reg.base + reg.nbytes:
                return reg, a - reg.base
        raise ValueError("all_to_all: `send` must be a tensor from this communicator's alloc() (peers read it in place)")

    def table(self, t, elem_offset=0):
        """Device array (int64) of the addresses of element `elem_offset` of tensor `t` (from alloc()) in every
        rank's copy: what a kernel needs to write into, or read from, the peers directly."""
        reg, off = self._region_of(t)
        return self._table(reg, off + int(elem_offset) * t.element_size())

    def alloc(self, shape, dtype):
        """COLLECTIVE.  A zero-filled tensor in exported memory."""
        n = 1
        for s in shape:
            n *= int(s)
        reg = self._new_region(max(1, n) * torch.empty(0, dtype=dtype).element_size())
        return reg.bytes[:n * torch.empty(0, dtype=dtype).element_size()].view(dtype).view(*shape)

    def free(self, t):
        """Releases the region behind a tensor from alloc() (call on every rank; `t` must not be used
        afterwards)."""
        reg, _ = self._region_of(t)
        torch.cuda.synchronize()
        dist.barrier(group=self.group)           # nobody is still reading it
        self._drop(reg)

    def _drop(self, reg):
        for a in reg.opened:
            H.peer_close(a)
        H.peer_free(reg.base)
        self._regions.remove(reg)
        for key in [k for k in self._tables if k[0] == id(reg)]:
            del self._tables[key]

    def close(self):
        torch.cuda.synchronize()
        dist.barrier(group=self.group)
        for reg in list(self._regions):
            self._drop(reg)
        self._stage = None

    # ------------------------------------------------------------------ stream-ordered collectives
    def barrier(self):
        H.peer_barrier(self._flags.base, self._flag_table, self._state, self.size, self.rank, self.timeout_s)
        self.barriers += 1

    def all_to_all(self, recv, send):
        """recv[j] = chunk `rank` of rank j's `send` (both [p, ...] contiguous, same bytes per chunk)."""
        if not send.is_contiguous() or send.numel() % self.size:
            raise ValueError("all_to_all: send must be contiguous and split into p equal chunks")
        reg, off = self._region_of(send)
        chunk = send.numel() * send.element_size() // self.size
        tab = self._table(reg, off)
        self.barrier()
        H.peer_all_to_all(recv, tab, chunk, self.size, self.rank)

    def all_reduce_sum(self, t):
        """In-place sum over the ranks of a contiguous fp32 tensor (added in rank order on every rank)."""
        if t.dtype != F32 or not t.is_contiguous():
            raise ValueError("all_reduce_sum: contiguous fp32 tensors only")
        nbytes = t.numel() * 4
        if self._stage is None or self._stage.nbytes < nbytes:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("first all_reduce_sum of this size under graph capture: run the step once eagerly")
            # a smaller staging region stays allocated and mapped until close(): graphs captured earlier hold
            # its addresses on this rank and on the peers
            self._stage = self._new_region(nbytes)
        tab = self._table(self._stage, 0)
        self._stage.bytes[:nbytes].view(F32).copy_(t.view(-1))
        self.barrier()                       # every rank has published its values
        H.peer_sum_f32(t, tab, self.size)
        self.barrier()                       # every rank has read them: the staging buffer is free again

    def status(self):
        """This rank's barrier status word (host synchronisation): 0, or 1 + j (rank j did not arrive in time at
        one of this rank's barriers), or 65 + j (rank j reported a failed barrier of its own)."""
        return int(self._state[1].item())

    def check(self, collective=True):
        """COLLECTIVE (host synchronisation; call it on every rank at the end of a run).  Raises on EVERY rank
        of the group if any rank's barrier gave up waiting: a rank that timed out went on unsynchronised, so the
        results of all ranks are invalid from that point, and a rank that raised alone would leave the others
        blocked in their next collective.  collective=False looks at this rank's status word only."""
        who = self.status()
        worst = who
        if collective and self.size > 1:
            from .host_staged import moves_host_memory_only
            t = torch.tensor([who], dtype=torch.int32, device="cpu" if moves_host_memory_only(self.group) else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            worst = int(t.item())
        if worst:
            if 1 <= who <= 64:
                mine = f"rank {who - 1} did not arrive at a barrier of rank {self.rank} within {self.timeout_s} s"
            elif who > 64:
                mine = f"rank {who - 65} reported a barrier that gave up waiting"
            else:
                mine = "another rank of the group reported a barrier that gave up waiting"
            raise RuntimeError(f"peer-mapped barrier failed: {mine}; results after that point are invalid on every rank")

    # ------------------------------------------------------------------ eager exchanges of the VAE decode
    def exchange_rows(self, up, down):
        return self.base.exchange_rows(up, down)

    def all_gather_rows(self, x, rows):
        return self.base.all_gather_rows(x, rows)
