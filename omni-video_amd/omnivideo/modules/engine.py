"""Execution plan for the DiT forward on MI355X.

One `DiTEngine` snapshots a WanModel's weights into the fused layouts the HIP kernels
want (QKV as one [3C,C] matrix, cross-attention K/V as one [2C,C] matrix), owns every
workspace buffer for a given (grid, seq_len, branches, sequence-parallel shard) shape, and
runs the forward as a fixed sequence of libomnihip launches on torch's current stream: no
allocation, no host synchronisation, so a whole sampling step can be captured into a HIP
graph.

Compared with the reference's eager module graph (omnivideo/modules/model.py:417-524)
the engine
  * evaluates the timestep MLP once per forward instead of once per token (the timestep
    is one scalar; model.py:467-476 expands it to [B, L]);
  * runs the conditional and unconditional branches of classifier-free guidance as one
    batch of 2 through every token-local kernel (x2x_gen_unified.py:480-481 runs two
    B=1 forwards);
  * embeds the context and projects the per-layer cross-attention K/V once per video
    (they do not depend on the latents or the timestep; model.py:170-171,480-485
    recompute them in every layer of every forward).
The arithmetic per element is unchanged, including the reference's bf16 rounding points.

Sequence parallelism (reference: omnivideo/distributed/full_sequence_parallel.py:63-232,
ulysses.py:8-56): rank r owns the contiguous token rows [r*S/p, (r+1)*S/p); everything
outside attention is token-local.  Self-attention exchanges q, k, v with ONE all-to-all
(sequence-gather / head-scatter, the three tensors packed into one message) and the
attention output with one more.  Cross-attention needs no exchange here: every rank keeps
the (small, step-invariant) context K/V of all heads and attends its own query rows,
where the reference shards the context and pays four more all-to-alls per layer
(full_sequence_parallel.py:221-227).  The head output is combined by one all-reduce of
the (zero-initialised) latent-sized result.
"""
import math

import torch
import torch.distributed as dist

from .. import _hip as H

BF16, F32 = torch.bfloat16, torch.float32


def build_rope_table(head_dim, rows, device, theta=10000.0):
    """float2 (cos, sin) [3, rows, head_dim/2]: axis a (frame, height, width) fills the
    complex pairs it owns inside a head; computed in float64 like the reference's
    rope_params (model.py:26-34,404-411) and rounded once to fp32."""
    c = head_dim // 2
    parts = (c - 2 * (c // 3), c // 3, c // 3)
    tab = torch.zeros(3, rows, c, 2, dtype=torch.float64)
    pos = torch.arange(rows, dtype=torch.float64)
    start = 0
    for a, npairs in enumerate(parts):
        dim_axis = 2 * npairs
        freq = 1.0 / torch.pow(theta, torch.arange(0, dim_axis, 2, dtype=torch.float64) / dim_axis)
        ang = pos[:, None] * freq[None, :]
        tab[a, :, start:start + npairs, 0] = ang.cos()
        tab[a, :, start:start + npairs, 1] = ang.sin()
        start += npairs
    return tab.to(F32).to(device).contiguous()


class ContextCache:
    """Embedded context and per-layer cross-attention K/V for `nb` guidance branches."""

    def __init__(self, kv, text_len, n_real=None):
        self.kv = kv            # [layers, nb, text_len, 2C] bf16: K (normalised) | V
        self.text_len = text_len
        # Rows [n_real, text_len) of every branch come from zero padding (model.py:480-485): the
        # same input row through the same per-row kernels, so their K and V rows are bit-identical.
        # The reference attends all of them (model.py:174, no k_lens); cross-attention reads keys
        # [0, n_real] and counts key n_real text_len - n_real times.
        n_real = text_len if n_real is None else min(int(n_real), text_len)
        self.k_len = min(n_real + 1, text_len)
        self.last_key_copies = text_len - self.k_len + 1


class TorchGroupComm:
    """The two collectives the sequence-parallel forward needs, over a torch.distributed
    group (backend "nccl" = RCCL over xGMI on MI355X)."""

    def __init__(self, group=None):
        self.group = group
        self.size = dist.get_world_size(group=group)
        self.rank = dist.get_rank(group=group)

    def all_to_all(self, recv, send):
        dist.all_to_all_single(recv, send, group=self.group)

    # split-phase form: the exchange is queued behind the work already on the current stream and runs on the
    # collective library's own stream; wait() makes the current stream wait for it.  What is launched between
    # the two runs beside the exchange.
    def all_to_all_start(self, recv, send, tag):
        if not hasattr(self, "_pending"):
            self._pending = {}
        self._pending[tag] = dist.all_to_all_single(recv, send, group=self.group, async_op=True)

    def wait(self, tag):
        self._pending.pop(tag).wait()

    def all_reduce_sum(self, t):
        dist.all_reduce(t, group=self.group)

    # --- the two exchanges of the row-sharded VAE decode (vae2_1.WanVAE_.decode with comm=...)
    def exchange_rows(self, up, down):
        """Sends `up` to rank - 1 and `down` to rank + 1; returns (row received from rank - 1, row
        received from rank + 1), None where there is no such neighbour."""
        ops, from_up, from_down = [], None, None
        g = self.group
        peer = (lambda r: dist.get_global_rank(g, r)) if g is not None else (lambda r: r)
        if self.rank > 0:
            from_up = torch.empty_like(up)
            ops += [dist.P2POp(dist.isend, up, peer(self.rank - 1), group=g),
                    dist.P2POp(dist.irecv, from_up, peer(self.rank - 1), group=g)]
        if self.rank < self.size - 1:
            from_down = torch.empty_like(down)
            ops += [dist.P2POp(dist.isend, down, peer(self.rank + 1), group=g),
                    dist.P2POp(dist.irecv, from_down, peer(self.rank + 1), group=g)]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return from_up, from_down

    def all_gather_rows(self, x, rows):
        """x [T, rows[rank], ...] on every rank -> [T, sum(rows), ...] on every rank."""
        mx = max(rows)
        pad = x.new_zeros((x.shape[0], mx) + tuple(x.shape[2:]))
        pad[:, :x.shape[1]] = x
        parts = [torch.empty_like(pad) for _ in range(self.size)]
        dist.all_gather(parts, pad, group=self.group)
        return torch.cat([p[:, :n] for p, n in zip(parts, rows)], dim=1)


def make_group_comm(group=None, kind=None):
    """The communicator for a sequence-parallel process group.  kind (default: OVD_SP_COMM, else "auto"):
      "rccl"  RCCL collectives on device tensors (the group's backend must be "nccl");
      "host"  the host-staged equivalent for a backend that only moves host memory (gloo: several ranks
              rehearsing on one GPU);
      "auto"  "host" on a gloo group, else "rccl";
      "peer"  peer-mapped device memory (distributed/peer_mapped.py): no collective library in the step,
              one graph per step; the group's own backend is used for set-up and for the VAE decode."""
    import os
    from ..distributed.host_staged import HostStagedComm, moves_host_memory_only
    kind = (kind or os.environ.get("OVD_SP_COMM") or "auto").lower()
    if kind == "auto":
        kind = "host" if moves_host_memory_only(group) else "rccl"
    if kind == "peer":
        from ..distributed.peer_mapped import PeerMappedComm
        return PeerMappedComm(group)
    if kind == "host":
        return HostStagedComm(group)
    if kind == "rccl":
        return TorchGroupComm(group)
    raise ValueError(f"OVD_SP_COMM / kind must be auto, rccl, host or peer, got {kind!r}")


FP8_LINEARS = ("qkv", "o", "cq", "co", "f0", "f2")


def planned_bytes(cfg, grid, seq_len, nb, text_len, sp=1, precision="bf16", attention=None, attn_workspace=0):
    """Device bytes one expert needs on one rank before anything is allocated: {"parameters": the module's
    bf16 parameters, "engine_copies": the fused / quantised weight copies DiTEngine.__init__ makes,
    "context_cache": encode_context's per-layer K/V for nb branches (+ its two transient embedding buffers),
    "workspaces": what plan() allocates for this shape (attn_workspace = H.flash_attn_workspace_bytes of the
    shape, which depends on the device), "total"}.  cfg: the WanModel configuration (dim, ffn_dim, num_heads,
    num_layers, text_dim, freq_dim, in_dim, out_dim, patch_size)."""
    C, F, N, nl = cfg["dim"], cfg["ffn_dim"], cfg["num_heads"], cfg["num_layers"]
    attention = attention or ("int8" if precision == "fp8" else "bf16")
    kpatch = cfg.get("in_dim", 16) * math.prod(cfg.get("patch_size", (1, 2, 2)))
    per_layer = 8 * C * C + 2 * C * F + 8 * C + C + F + 2 * C + 4 * C + 6 * C    # linears, biases, norms, modulation
    outside = (kpatch * C + C) + (cfg["text_dim"] * C + C * C + 2 * C) + (cfg.get("freq_dim", 256) * C + C * C + 2 * C) \
        + (6 * C * C + 6 * C) + (C * cfg.get("out_dim", 16) * math.prod(cfg.get("patch_size", (1, 2, 2))) + 2 * C + 64)
    params = 2 * (nl * per_layer + outside)
    if precision == "fp8":
        copies = nl * (2 * (2 * C * C + 2 * C) + (6 * C * C + 2 * C * F) + 4 * (8 * C + 2 * F))
    else:
        copies = nl * 2 * (5 * C * C + 5 * C)
    ctx = nl * nb * text_len * 2 * C * 2 + 2 * nb * text_len * C * 2 + nb * text_len * cfg["text_dim"] * 2
    S = seq_len // sp
    M = nb * S
    n_tok = grid[0] * grid[1] * grid[2]
    ws = M * C * 4 + M * 3 * C * 2 + M * C * 2 + M * C * 2 + M * C * 4 + n_tok * kpatch * 2     # x, qkv, att, res, y, rows
    ws += (2 * M * C + 8 * M + M * F) if precision == "fp8" else (M * C * 2 + M * F * 2)        # h / att / ffn operands
    ws += nl * 6 * C * 4 + attn_workspace
    if attention == "int8":
        ws += 2 * nb * seq_len * (C // sp) + 4 * nb * seq_len * (N // sp) + 4 * nb * -(-seq_len // 64) * (N // sp)
    if sp > 1:
        ws += (3 + 1 + 3 + 1) * nb * seq_len * (C // sp) * 2      # send image / operands / output / returned output
    return {"parameters": params, "engine_copies": copies, "context_cache": ctx, "workspaces": ws,
            "total": params + copies + ctx + ws}



class DiTEngine:
    def __init__(self, model, device=None, precision="bf16", attention=None):
        """model: omnivideo.modules.model.WanModel (parameters are read, never modified).
        precision: "bf16" (the reference's arithmetic) or "fp8": the six large linear layers of
        every block run on e4m3 operands - weights quantised once with a scale per output
        channel, activations per token row inside the kernel that produces them (the GELU output
        with the fixed scale `ffn_act_scale`).
        attention: "bf16", or "int8": self-attention computes Q K^T on int8 operands (q with a scale per
        token and head, k per 64-key tile and head; softmax, P and P V unchanged); default: "int8"
        with precision "fp8", else "bf16".  Cross-attention, norms, the residual stream and the head
        are the same in every mode."""
        if precision not in ("bf16", "fp8"):
            raise ValueError(f"precision must be 'bf16' or 'fp8', got {precision!r}")
        attention = attention or ("int8" if precision == "fp8" else "bf16")
        if attention not in ("bf16", "int8"):
            raise ValueError(f"attention must be 'bf16' or 'int8', got {attention!r}")
        self.precision, self.attention = precision, attention
        self.ffn_act_scale = 1.0
        p0 = model.patch_embedding.weight
        self.device = torch.device(device) if device is not None else p0.device
        if self.device.type != "cuda":
            raise RuntimeError("DiTEngine needs the model on a HIP device; there is no CPU path")
        self.dim, self.ffn_dim = model.dim, model.ffn_dim
        self.num_heads, self.num_layers = model.num_heads, model.num_layers
        self.head_dim = self.dim // self.num_heads
        self.eps, self.freq_dim = model.eps, model.freq_dim
        self.patch_size, self.in_dim, self.out_dim = tuple(model.patch_size), model.in_dim, model.out_dim
        self.text_dim = model.text_dim
        if self.head_dim != 128:
            raise RuntimeError(f"head_dim {self.head_dim} unsupported: the attention kernel is built for 128")

        def w(t):
            return t.detach().to(self.device, BF16).contiguous()

        self.w_patch = w(p0).reshape(self.dim, -1)
        self.b_patch = w(model.patch_embedding.bias)
        te, tm, tp = model.text_embedding, model.time_embedding, model.time_projection
        self.w_te0, self.b_te0, self.w_te2, self.b_te2 = w(te[0].weight), w(te[0].bias), w(te[2].weight), w(te[2].bias)
        self.w_tm0, self.b_tm0, self.w_tm2, self.b_tm2 = w(tm[0].weight), w(tm[0].bias), w(tm[2].weight), w(tm[2].bias)
        self.w_tp, self.b_tp = w(tp[1].weight), w(tp[1].bias)
        self.layers = []
        mods = []
        for blk in model.blocks:
            sa, ca = blk.self_attn, blk.cross_attn
            L = dict(
                w_qkv=torch.cat([w(sa.q.weight), w(sa.k.weight), w(sa.v.weight)], 0).contiguous(),
                b_qkv=torch.cat([w(sa.q.bias), w(sa.k.bias), w(sa.v.bias)], 0).contiguous(),
                nq=w(sa.norm_q.weight), nk=w(sa.norm_k.weight), w_o=w(sa.o.weight), b_o=w(sa.o.bias),
                n3w=w(blk.norm3.weight), n3b=w(blk.norm3.bias),
                w_cq=w(ca.q.weight), b_cq=w(ca.q.bias), cnq=w(ca.norm_q.weight), cnk=w(ca.norm_k.weight),
                w_ckv=torch.cat([w(ca.k.weight), w(ca.v.weight)], 0).contiguous(),
                b_ckv=torch.cat([w(ca.k.bias), w(ca.v.bias)], 0).contiguous(),
                w_co=w(ca.o.weight), b_co=w(ca.o.bias),
                w_f0=w(blk.ffn[0].weight), b_f0=w(blk.ffn[0].bias), w_f2=w(blk.ffn[2].weight), b_f2=w(blk.ffn[2].bias),
                mod=w(blk.modulation).reshape(6, self.dim))
            if precision == "fp8":
                for name in FP8_LINEARS:
                    L["w8_" + name], L["s_" + name] = H.quant_rows_fp8(L["w_" + name])
                    L["w_" + name] = None            # the bf16 copy is not read again
            self.layers.append(L)
            mods.append(L["mod"].data_ptr())
        self.mod_ptrs = torch.tensor(mods, dtype=torch.int64, device=self.device)
        self.w_head, self.b_head = w(model.head.head.weight), w(model.head.head.bias)
        self.head_mod = w(model.head.modulation).reshape(2, self.dim)
        self._shape = None
        self.timing = None
        import os
        self.sp_overlap = os.environ.get("OVD_SP_OVERLAP", "1") != "0"

    # ------------------------------------------------------------------ workspaces
    def plan(self, grid, seq_len, nb, comm=None):
        """Allocates every buffer for `nb` guidance branches over a `seq_len`-row sequence.
        comm: a communicator (size, rank, all_to_all, all_reduce_sum) for sequence
        parallelism; this rank then owns rows [rank*seq_len/size, (rank+1)*seq_len/size)."""
        sp = 1 if comm is None else comm.size
        rank = 0 if comm is None else comm.rank
        key = (tuple(grid), int(seq_len), int(nb), sp, rank, id(comm))
        if self._shape == key:
            return
        dev, C, F, N = self.device, self.dim, self.ffn_dim, self.num_heads
        if seq_len % sp:
            raise ValueError(f"seq_len {seq_len} must be a multiple of the sequence-parallel size {sp}")
        if N % sp:
            raise ValueError(f"q.num_heads={N} must be divisible by sp_size={sp}")   # reference ulysses.py:33-34
        S = seq_len // sp
        M = nb * S
        n_tok = grid[0] * grid[1] * grid[2]
        if n_tok > seq_len:
            raise ValueError(f"seq_len {seq_len} shorter than the {n_tok} latent tokens")
        kpatch = self.in_dim * math.prod(self.patch_size)
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        self.x = e(nb, S, C, dt=F32)
        self.qkv = e(nb, S, 3 * C)
        self.att = e(nb, S, C)
        self.res = e(M, C)             # output of a sub-layer's last Linear until the next norm adds it to x
        if self.precision == "fp8":
            self.h8, self.h_scale = e(M, C, dt=torch.uint8), e(M, dt=F32)
            self.att8, self.att_scale = e(M, C, dt=torch.uint8), e(M, dt=F32)
            self.ffn8 = e(M, F, dt=torch.uint8)
            self.h = self.ffn = None
        else:
            self.h = e(M, C)
            self.ffn = e(M, F)
        self.y = e(M, C, dt=F32)
        self.rows = e(n_tok, kpatch)
        self.sin = e(self.freq_dim, dt=F32)
        self.e_tmp, self.e_vec = e(C, dt=F32), e(C, dt=F32)
        self.e0 = e(6 * C, dt=F32)
        self.mod = e(self.num_layers, 6, C, dt=F32)
        self.hmod = e(2, C, dt=F32)
        self.rope = build_rope_table(self.head_dim, max(grid), dev)
        # self-attention runs over the full sequence for N / sp heads: workspace for its tail balancing
        ws = H.flash_attn_workspace_bytes(nb, seq_len, N // sp, n_tok)
        if sp > 1:      # the sharded forward may launch self-attention one guidance branch at a time
            ws = max(ws, H.flash_attn_workspace_bytes(1, seq_len, N // sp, n_tok))
        self.attn_ws = torch.empty(ws, dtype=torch.uint8, device=dev) if ws > 0 else None
        if self.attention == "int8":         # self-attention operands: full sequence, this rank's heads
            self.q8, self.k8 = e(nb, seq_len, C // sp, dt=torch.int8), e(nb, seq_len, C // sp, dt=torch.int8)
            self.q8_scale = e(nb, seq_len, N // sp, dt=F32)
            self.k8_scale = e(nb, -(-seq_len // 64), N // sp, dt=F32)
        if sp > 1:
            Cs = C // sp
            # what the peers receive from: ordinary memory for a collective library, exported memory for the
            # peer-mapped communicator (its ranks read these two buffers in place)
            shared = getattr(comm, "alloc", None)
            self.sp_direct = shared is not None and bool(getattr(comm, "direct", False))
            if shared is not None and getattr(self, "comm", None) is comm:
                for old in getattr(self, "_sp_shared", ()):
                    comm.free(old)
            ex = (lambda *s: shared(s, BF16)) if shared is not None else e
            if self.sp_direct:
                # the peers write q|k|v into sp_qkv and read sp_o themselves: no send image, no receive copy
                self.sp_qkv = ex(3, nb, seq_len, Cs)
                self.sp_o = ex(nb, seq_len, Cs)
                self.sp_send = self.sp_orecv = None
                self._sp_shared = (self.sp_qkv, self.sp_o)
                row0_el = rank * S * Cs
                self._sp_qkv_tab = comm.table(self.sp_qkv, row0_el)
                self._sp_qkv_tabs = [comm.table(self.sp_qkv, b * seq_len * Cs + row0_el) for b in range(nb)]
                self._sp_o_tabs = [comm.table(self.sp_o, b * seq_len * Cs + row0_el) for b in range(nb)]
            else:
                self.sp_send = ex(3, nb, sp, S, Cs)   # q|k|v of my rows, split by destination head group
                self.sp_qkv = e(3, nb, seq_len, Cs)   # the full sequence for my head group (received in place)
                self.sp_o = ex(nb, seq_len, Cs)       # attention output, full sequence, my head group
                self.sp_orecv = e(nb, sp, S, Cs)      # every head group's output for my rows
                self._sp_shared = (self.sp_send, self.sp_o) if shared is not None else ()
        if sp > 1 and getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=dev)       # carries one branch's exchange beside the other's attention
        self.grid, self.seq_len, self.nb, self.n_tok = tuple(grid), seq_len, nb, n_tok
        self.comm, self.sp, self.S, self.row0 = comm, sp, S, rank * S
        self._shape = key

    # ------------------------------------------------------------------ context (once per video)
    def encode_context(self, contexts, text_len):
        """contexts: list (one per guidance branch) of [n_i, text_dim] tensors, n_i <= text_len.
        Zero-pads to text_len, runs text_embedding (model.py:480-485) and every layer's
        cross-attention K projection + RMSNorm and V projection (model.py:170-171)."""
        nb, C, dev = len(contexts), self.dim, self.device
        pad = torch.zeros(nb, text_len, self.text_dim, dtype=BF16, device=dev)
        for i, c in enumerate(contexts):
            c = c.reshape(-1, self.text_dim)
            if c.shape[0] > text_len:
                raise ValueError(f"context of {c.shape[0]} rows exceeds text_len {text_len}")
            pad[i, :c.shape[0]] = c.to(dev, BF16)
        hid = torch.empty(nb * text_len, C, dtype=BF16, device=dev)
        emb = torch.empty(nb * text_len, C, dtype=BF16, device=dev)
        H.gemm(pad.view(nb * text_len, -1), self.w_te0, hid, self.b_te0, H.EPI_BF16_GELU)
        H.gemm(hid, self.w_te2, emb, self.b_te2, H.EPI_BF16)
        kv = torch.empty(self.num_layers, nb, text_len, 2 * C, dtype=BF16, device=dev)
        for l, L in enumerate(self.layers):
            flat = kv[l].view(nb * text_len, 2 * C)
            H.gemm(emb, L["w_ckv"], flat, L["b_ckv"], H.EPI_BF16)
            kview = flat[:, :C]
            H.rmsnorm_rope(kview, L["cnk"], kview, self.head_dim, self.eps)
        return ContextCache(kv, text_len, max(int(c.reshape(-1, self.text_dim).shape[0]) for c in contexts))

    # ------------------------------------------------------------------ attention variants
    def _self_attention(self, q4, k4, v4, att4, branch=None):
        """branch: None = all guidance branches in one launch; b = branch b alone."""
        int8 = self.attention == "int8"
        sl = slice(None) if branch is None else slice(branch, branch + 1)
        q4, k4, v4, att4 = q4[sl], k4[sl], v4[sl], att4[sl]
        if int8:
            q8, k8 = (t[sl].unflatten(2, (q4.shape[2], q4.shape[3])) for t in (self.q8, self.k8))
            q8s, k8s = self.q8_scale[sl], self.k8_scale[sl]
            H.quant_heads_i8(q4, q8, q8s, 1)
            H.quant_heads_i8(k4, k8, k8s, 64)
        if self.timing is not None:      # profiling hook: events around the dominant kernel
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if int8:
            H.flash_attn_qk8(q8, q8s, k8, k8s, v4, att4, k_len=self.n_tok, workspace=self.attn_ws)
        else:
            H.flash_attn(q4, k4, v4, att4, k_len=self.n_tok, workspace=self.attn_ws)
        if self.timing is not None:
            ev[1].record()
            self.timing.append(ev + (q4.shape[0],))          # (start, end, guidance branches in the launch)

    def _self_attention_sp(self, L):
        """Ulysses exchange around the attention kernel (reference ulysses.py:41-55).  The kernel that
        normalises and rotates q and k writes q, k and v straight into the all-to-all send image (split
        by destination head group); each (tensor, guidance branch) slab is exchanged on its own so that
        what arrives IS the full sequence for this rank's heads, which the attention kernel reads in
        place; its output goes back the same way and one gather kernel restores token rows.  No packing
        or unpacking copies (the reference chunks, copies and concatenates around each of its four
        all-to-alls, util.py:106-109).

        The two guidance branches are independent until the guidance combine, so with `overlap` (default;
        OVD_SP_OVERLAP=0 restores the serial order) attention runs one branch at a time and the other branch's
        exchange runs beside it: the second branch's q | k | v travel while the first branch attends, the
        first branch's output returns while the second attends.  The reference issues everything in sequence
        on one stream (ulysses.py:41-55)."""
        N, D, nb, S, sp = self.num_heads, self.head_dim, self.nb, self.S, self.sp
        overlap = nb == 2 and self.sp_overlap
        q4, k4, v4 = (self.sp_qkv[i].unflatten(2, (N // sp, D)) for i in range(3))
        o4 = self.sp_o.unflatten(2, (N // sp, D))
        if self.sp_direct and overlap:
            # Peer-mapped, overlapped.  Three barriers per layer on the main stream; a barrier vouches for what its
            # rank issued on the main stream before it, so work on the side stream is joined before the barrier
            # that has to cover it.
            #   B1: every rank's q | k | v of branch 0 have landed               -> attention of branch 0
            #   B2: ... attention of branch 0 is done, q | k | v of branch 1 have landed -> attention of branch 1,
            #       and (side stream) the gather of branch 0's output from the peers
            #   B3: ... attention of branch 1 is done                            -> gather of branch 1
            # Overwrites: a peer's q | k | v of branch b are rewritten (next layer) after B2 / B3 of this layer,
            # i.e. after that peer's attention of branch b; a peer's output of branch 0 is rewritten by its next
            # attention of branch 0, which follows its B1 of the next layer, which waits for this rank's B1, which
            # this rank issues after joining its side stream (gather of branch 0 done).
            main, side = torch.cuda.current_stream(), self._side
            qkv_b = [self.qkv[b] for b in range(nb)]
            args = (L["nq"], L["nk"], D, self.eps, self.rope, S, self.row0, self.n_tok, self.grid)
            H.qkv_norm_rope_scatter(qkv_b[0], *args, self._sp_qkv_tabs[0], self.seq_len, samples_total=nb)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                H.qkv_norm_rope_scatter(qkv_b[1], *args, self._sp_qkv_tabs[1], self.seq_len, samples_total=nb)
            self.comm.barrier()                                   # B1
            self._self_attention(q4, k4, v4, o4, branch=0)
            main.wait_stream(side)
            self.comm.barrier()                                   # B2
            side.wait_stream(main)
            with torch.cuda.stream(side):
                H.gather_col_blocks_from(self._sp_o_tabs[0], self.att[0], first=self.comm.rank + 1)
            self._self_attention(q4, k4, v4, o4, branch=1)
            self.comm.barrier()                                   # B3
            H.gather_col_blocks_from(self._sp_o_tabs[1], self.att[1], first=self.comm.rank + 1)
            main.wait_stream(side)
            return
        if self.sp_direct:
            # Peer-mapped exchange: q, k, v go from the kernel that produces them into every peer's attention
            # operands; after the barrier all of mine have arrived.  The second barrier says every rank's
            # attention is done: the outputs may be read from the peers (and sp_qkv may be written again).
            H.qkv_norm_rope_scatter(self.qkv.view(nb * S, -1), L["nq"], L["nk"], D, self.eps, self.rope, S, self.row0,
                                    self.n_tok, self.grid, self._sp_qkv_tab, self.seq_len)
            self.comm.barrier()
            self._self_attention(q4, k4, v4, o4)
            self.comm.barrier()
            for b in range(nb):
                H.gather_col_blocks_from(self._sp_o_tabs[b], self.att[b], first=self.comm.rank + 1)
            return
        H.qkv_norm_rope(self.qkv.view(nb * S, -1), L["nq"], L["nk"], D, self.eps, self.rope, S, self.row0,
                        self.n_tok, self.grid, send=self.sp_send)
        if overlap and hasattr(self.comm, "all_to_all_start"):
            # Collective library, split-phase: all six q | k | v exchanges are queued at once (they run in this
            # order on the library's stream); attention of branch 0 needs the first three only.
            for b in range(nb):
                for part in range(3):
                    self.comm.all_to_all_start(self.sp_qkv[part, b].view(sp, S, -1), self.sp_send[part, b], ("qkv", b, part))
            for b in range(nb):
                for part in range(3):
                    self.comm.wait(("qkv", b, part))
                self._self_attention(q4, k4, v4, o4, branch=b)
                self.comm.all_to_all_start(self.sp_orecv[b], self.sp_o[b].view(sp, S, -1), ("o", b))
            for b in range(nb):
                self.comm.wait(("o", b))
                H.gather_col_blocks(self.sp_orecv[b], self.att[b])
            return
        for part in range(3):
            for b in range(nb):
                self.comm.all_to_all(self.sp_qkv[part, b].view(sp, S, -1), self.sp_send[part, b])
        self._self_attention(q4, k4, v4, o4)
        for b in range(nb):
            self.comm.all_to_all(self.sp_orecv[b], self.sp_o[b].view(sp, S, -1))
            H.gather_col_blocks(self.sp_orecv[b], self.att[b])

    # ------------------------------------------------------------------ sub-layer pieces
    def _norm(self, x2, pending, **kw):
        """LayerNorm (+ affine / modulation) of the residual stream into the GEMM input buffer (bf16
        `h`, or e4m3 `h8` + row scales), after adding the pending residual update (y, gate) to x."""
        fp8 = self.precision == "fp8"
        if pending is not None:
            H.resid_layernorm_modulate(x2, pending[0], self.h8 if fp8 else self.h, self.eps, res_gate=pending[1],
                                       row_scale=self.h_scale if fp8 else None, **kw)
        elif fp8:
            H.layernorm_modulate_fp8(x2, self.h8, self.h_scale, self.eps, **kw)
        else:
            H.layernorm_modulate(x2, self.h, self.eps, **kw)

    def _linear_from_attention(self, att2, L, name):
        """self.res = Linear(attention output) in bf16 (the o projections, model.py:152, 177)."""
        if self.precision == "fp8":
            H.quant_rows_fp8(att2, self.att8, self.att_scale)
            H.gemm_fp8(self.att8, L["w8_" + name], L["s_" + name], self.res, L["b_" + name], H.EPI_BF16,
                       a_scale=self.att_scale)
        else:
            H.gemm(att2, L["w_" + name], self.res, L["b_" + name], H.EPI_BF16)

    # ------------------------------------------------------------------ forward
    def forward(self, lat, timesteps, ctx, out, step=0, step_ptr=None):
        """lat [Cin,T,H,W] (bf16 or fp32, rounded to bf16 on entry as the reference's bf16
        latents are), timesteps float64 device table, ctx a ContextCache for self.nb branches,
        out fp32 [nb, Cout, T, H, W].  plan() must have been called for this shape.  Under
        sequence parallelism every rank passes the same lat / ctx and receives the full out."""
        C, N, D, nb, S = self.dim, self.num_heads, self.head_dim, self.nb, self.S
        M = nb * S
        n_tok, row0 = self.n_tok, self.row0
        n_loc = max(0, min(S, n_tok - row0))          # real (non-padding) tokens among my rows
        if ctx.kv.shape[1] != nb:
            raise ValueError("context cache was built for a different number of branches")
        # patch embedding (model.py:455-464): rows beyond the real tokens stay zero
        H.patchify(lat, self.rows, self.patch_size)
        self.x.zero_()
        if n_loc > 0:
            H.gemm(self.rows[row0:row0 + n_loc], self.w_patch, self.x[0, :n_loc], self.b_patch, H.EPI_F32)
        for b in range(1, nb):
            self.x[b].copy_(self.x[0])
        # timestep embedding, once (model.py:467-476)
        H.time_sinusoid(timesteps, self.sin, step, step_ptr)
        H.gemv_f32(self.sin, self.w_tm0, self.b_tm0, self.e_tmp)
        H.gemv_f32(self.e_tmp, self.w_tm2, self.b_tm2, self.e_vec, silu_in=True)
        H.gemv_f32(self.e_vec, self.w_tp, self.b_tp, self.e0, silu_in=True)
        H.modulation_table(self.mod_ptrs, self.e0, self.mod)
        x2 = self.x.view(M, C)
        qkv2 = self.qkv.view(M, 3 * C)
        q4 = self.qkv[:, :, :C].unflatten(2, (N, D))
        k4 = self.qkv[:, :, C:2 * C].unflatten(2, (N, D))
        v4 = self.qkv[:, :, 2 * C:].unflatten(2, (N, D))
        att4 = self.att.unflatten(2, (N, D))
        att2 = self.att.view(M, C)
        fp8 = self.precision == "fp8"
        # The residual updates x = x + y * e (model.py:245-246, 250, 253-254) are applied by the norm
        # kernel that follows each of them (resid_layernorm_modulate): `pending` carries (y, e).
        pending = None
        for l, L in enumerate(self.layers):
            mod = self.mod[l]
            # self-attention (model.py:242-246)
            self._norm(x2, pending, shift=mod[0], scale=mod[1], round_norm=(l == 0))
            if fp8:
                H.gemm_fp8(self.h8, L["w8_qkv"], L["s_qkv"], qkv2, L["b_qkv"], H.EPI_BF16, a_scale=self.h_scale)
            else:
                H.gemm(self.h, L["w_qkv"], qkv2, L["b_qkv"], H.EPI_BF16)
            if self.sp > 1:
                self._self_attention_sp(L)
            else:
                H.qkv_norm_rope(qkv2, L["nq"], L["nk"], D, self.eps, self.rope, S, row0, n_tok, self.grid)
                self._self_attention(q4, k4, v4, att4)
            self._linear_from_attention(att2, L, "o")
            pending = (self.res, mod[2])
            # cross-attention (model.py:250): local query rows against the replicated context K/V
            self._norm(x2, pending, weight=L["n3w"], bias=L["n3b"])
            qc = qkv2[:, :C]
            if fp8:
                H.gemm_fp8(self.h8, L["w8_cq"], L["s_cq"], qc, L["b_cq"], H.EPI_BF16, a_scale=self.h_scale)
            else:
                H.gemm(self.h, L["w_cq"], qc, L["b_cq"], H.EPI_BF16)
            H.rmsnorm_rope(qc, L["cnq"], qc, D, self.eps)
            kvl = ctx.kv[l]
            # (the workspace only matters for long contexts - V2V with thousands of visual tokens - and
            # for the unsharded case: it is sized for all N heads over seq_len query rows)
            H.flash_attn(q4, kvl[:, :, :C].unflatten(2, (N, D)), kvl[:, :, C:].unflatten(2, (N, D)), att4,
                         k_len=ctx.k_len, last_key_copies=ctx.last_key_copies,
                         workspace=self.attn_ws if self.sp == 1 else None)
            self._linear_from_attention(att2, L, "co")
            pending = (self.res, None)
            # feed-forward (model.py:251-254)
            self._norm(x2, pending, shift=mod[3], scale=mod[4])
            if fp8:
                H.gemm_fp8(self.h8, L["w8_f0"], L["s_f0"], self.ffn8, L["b_f0"], H.EPI_FP8_GELU, a_scale=self.h_scale,
                           out_scale=self.ffn_act_scale)
                H.gemm_fp8(self.ffn8, L["w8_f2"], L["s_f2"], self.res, L["b_f2"], H.EPI_BF16,
                           a_scale_all=self.ffn_act_scale)
            else:
                H.gemm(self.h, L["w_f0"], self.ffn, L["b_f0"], H.EPI_BF16_GELU)
                H.gemm(self.ffn, L["w_f2"], self.res, L["b_f2"], H.EPI_BF16)
            pending = (self.res, mod[5])
        # head + unpatchify (model.py:520-524)
        H.head_modulation(self.head_mod, self.e_vec, self.hmod)
        if pending is None:
            H.layernorm_modulate(x2, self.y, self.eps, shift=self.hmod[0], scale=self.hmod[1])
        else:
            H.resid_layernorm_modulate(x2, pending[0], self.y, self.eps, res_gate=pending[1], shift=self.hmod[0],
                                       scale=self.hmod[1])
        if self.sp > 1:
            out.zero_()
        for b in range(nb):
            H.head_unpatchify(self.y[b * S:(b + 1) * S], self.w_head, self.b_head, out[b], self.grid,
                              self.patch_size, self.out_dim, tok_offset=row0)
        if self.sp > 1:
            # every latent element has exactly one owner rank: the sum is an exact gather
            # (the reference all-gathers the head output rows, full_sequence_parallel.py:163)
            self.comm.all_reduce_sum(out)
        return out
