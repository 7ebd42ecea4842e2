"""ctypes binding of libomnihip.so (the C-ABI declared in include/omnihip.h).

There is no fallback: if the library is missing or an entry point is rejected the call
raises.  Tensors are borrowed for the duration of the call; every launch goes to torch's
current HIP stream so the calls compose with torch ops and can be graph-captured.
"""
import ctypes
import os

import torch

# OVD_LIB_PATH: load another build of the library (kernel-variant timing; the default is the in-tree build)
_LIB_PATH = os.environ.get("OVD_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libomnihip.so")
_lib = None

_i, _l, _f, _p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

SIGNATURES = {
    "ovd_layernorm_modulate_bf16": [_p, _l, _p, _p, _p, _p, _p, _l, _i, _i, _f, _i, _p],
    "ovd_layernorm_modulate_f32": [_p, _l, _p, _p, _p, _p, _p, _l, _i, _i, _f, _i, _p],
    "ovd_rmsnorm_rope_bf16": [_p, _l, _p, _p, _i, _p, _l, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _p],
    "ovd_qkv_norm_rope_bf16": [_p, _l, _p, _p, _p, _i, _p, _l, _l, _i, _l, _l, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _p],
    "ovd_gather_col_blocks_bf16": [_p, _p, _l, _l, _i, _i, _p],
    "ovd_qkv_norm_rope_scatter_bf16": [_p, _l, _p, _p, _p, _i, _p, _l, _l, _i, _l, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _p],
    "ovd_gather_col_blocks_from_bf16": [_p, _p, _l, _l, _i, _i, _i, _p],
    "ovd_gemm_bf16": [_p, _l, _p, _l, _p, _p, _l, _p, _i, _i, _i, _i, _p],
    "ovd_flash_attn_fwd_bf16": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _l, _l, _l, _l, _f, _p],
    "ovd_flash_attn_tail_fwd_bf16": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _l, _l, _l, _l, _f, _p],
    "ovd_flash_attn_ws_fwd_bf16": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _l, _l, _l, _l, _f, _p,
                                   _l, _p],
    "ovd_flash_attn_workspace_bytes": [_i, _i, _i, _i],
    "ovd_flash_attn_static_fwd_bf16": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _l, _l, _l, _l, _f,
                                       _p],
    "ovd_quant_heads_i8": [_p, _l, _l, _p, _l, _l, _p, _i, _i, _i, _i, _i, _p],
    "ovd_flash_attn_qk8_fwd_bf16": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _l, _l, _l, _l, _f,
                                    _p, _l, _p],
    "ovd_patchify": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ovd_time_sinusoid": [_p, _p, _i, _p, _i, _p],
    "ovd_gemv_f32": [_p, _p, _p, _p, _i, _i, _i, _p],
    "ovd_modulation_table": [_p, _p, _p, _i, _i, _p],
    "ovd_head_modulation": [_p, _p, _p, _i, _p],
    "ovd_head_unpatchify": [_p, _l, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "ovd_cfg_unipc_step": [_p, _p, _p, _p, _p, _p, _p, _f, _p, _i, _l, _i, _p],
    "ovd_cfg_dpm_step": [_p, _p, _p, _p, _p, _p, _f, _p, _i, _l, _i, _p],
    "ovd_video_to_uint8": [_p, _p, _l, _l, _i, _l, _f, _f, _p],
    "ovd_resid_layernorm_modulate": [_p, _l, _p, _l, _p, _p, _p, _p, _p, _p, _l, _i, _p, _i, _i, _f, _i, _p],
    "ovd_quant_rows_fp8": [_p, _l, _p, _l, _p, _i, _i, _p],
    "ovd_layernorm_modulate_fp8": [_p, _l, _p, _p, _p, _p, _p, _l, _p, _i, _i, _f, _i, _p],
    "ovd_gemm_fp8": [_p, _l, _p, _f, _p, _l, _p, _p, _p, _l, _p, _f, _i, _i, _i, _i, _p],
    "ovd_resize_coeffs": [_i, _i, _p, _p, _i],
    "ovd_frames_resize_normalize": [_p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "ovd_t5_attention_bf16": [_p, _l, _p, _p, _p, _p, _l, _i, _i, _i, _i, _p],
    "ovd_t5_gated_gelu_bf16": [_p, _l, _p, _l, _i, _i, _p],
    "ovd_add_bf16": [_p, _p, _l, _p],
    "ovd_step_advance": [_p, _p],
    "ovd_vae_conv3d_f32": [_p] * 7 + [_i] + [_p] * 4 + [_i] * 14 + [_p],
    "ovd_vae_conv3d_sw_f32": [_p] * 7 + [_i] + [_p] * 4 + [_i] * 14 + [_p],
    "ovd_vae_split_weights": [_p, _p, _l, _i, _p],
    "ovd_vae_act_split": [_p, _p, _i, _p, _l, _i, _p],
    "ovd_vae_conv3d_ps_f32": [_p] * 7 + [_i] * 11 + [_p],
    "ovd_vae_voxel_rnorm": [_p, _p, _l, _i, _p],
    "ovd_vae_to_channels_last": [_p, _p, _p, _p, _p, _i, _i, _l, _p],
    "ovd_vae_from_channels_last": [_p, _p, _p, _p, _p, _i, _i, _l, _i, _p],
    "ovd_softmax_rows_f32": [_p, _i, _i, _f, _p],
    "ovd_peer_alloc": [_p, _l, _i],
    "ovd_peer_free": [_p],
    "ovd_peer_export": [_p, _p],
    "ovd_peer_open": [_p, _p],
    "ovd_peer_close": [_p],
    "ovd_peer_barrier": [_p, _p, _p, _i, _i, ctypes.c_double, _p],
    "ovd_peer_all_to_all": [_p, _p, _l, _i, _i, _p],
    "ovd_peer_sum_f32": [_p, _p, _l, _i, _p],
}

EPI_BF16, EPI_BF16_GELU, EPI_F32, EPI_RESID = 0, 1, 2, 3
UNIPC_COEF_STRIDE = 16
DPM_COEF_STRIDE = 8


def lib():
    """Loads libomnihip.so once; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} not found: build it with `make -C omni-video_amd/csrc` "
                "(or __graft_entry__.build()); there is no non-HIP fallback")
        L = ctypes.CDLL(_LIB_PATH)
        L.ovd_last_error.restype = ctypes.c_char_p
        L.ovd_abi_version.restype = _i
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _l if name == "ovd_flash_attn_workspace_bytes" else _i
        _lib = L
    return _lib


CALLS = 0          # entry-point calls made by this process (segmented graph capture counts launches with it)


def _call(name, *args):
    global CALLS
    L = lib()
    CALLS += 1
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name}: {L.ovd_last_error().decode()}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t, dtype=None, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError("tensor argument is None")
    if not t.is_cuda:
        raise ValueError("libomnihip works on device tensors only (got a CPU tensor)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def _rows(t):
    """(M, C, row stride) of a tensor whose last dim is contiguous and whose leading dims
    collapse to one uniform row stride."""
    if t.stride(-1) != 1:
        raise ValueError("last dimension must be contiguous")
    C = t.shape[-1]
    M = t.numel() // C
    if t.dim() == 1:
        return 1, C, C
    ld = t.stride(-2)
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            raise ValueError("leading dimensions do not collapse to a uniform row stride")
        exp *= t.shape[d]
    return M, C, ld


BF16, F32 = torch.bfloat16, torch.float32


def layernorm_modulate(x, out, eps, shift=None, scale=None, weight=None, bias=None, round_norm=False):
    M, C, ldx = _rows(x)
    Mo, Co, ldo = _rows(out)
    if (Mo, Co) != (M, C):
        raise ValueError("layernorm: out shape mismatch")
    for v in (shift, scale):
        if v is not None and (v.numel() != C or not v.is_contiguous()):
            raise ValueError("layernorm: shift/scale must be contiguous rows of length C")
    name = "ovd_layernorm_modulate_bf16" if out.dtype == BF16 else "ovd_layernorm_modulate_f32"
    if out.dtype not in (BF16, F32):
        raise TypeError("layernorm: out must be bf16 or fp32")
    _call(name, _ptr(x, F32), ldx, _ptr(shift, F32, True), _ptr(scale, F32, True), _ptr(weight, BF16, True),
          _ptr(bias, BF16, True), _ptr(out), ldo, M, C, float(eps), int(round_norm), _stream())
    return out


def resid_layernorm_modulate(x, res, out, eps, res_gate=None, shift=None, scale=None, weight=None, bias=None,
                             round_norm=False, row_scale=None):
    """x += res * res_gate (in place), then LayerNorm (+ affine) (+ modulation) of the updated x into
    out (bf16, fp32, or uint8 = e4m3 with row_scale)."""
    M, C, ldx = _rows(x)
    Mr, Cr, ldr = _rows(res)
    Mo, Co, ldo = _rows(out)
    if (Mo, Co) != (M, C) or (Mr, Cr) != (M, C):
        raise ValueError("layernorm: shape mismatch")
    for v in (shift, scale, res_gate):
        if v is not None and (v.numel() != C or not v.is_contiguous()):
            raise ValueError("layernorm: shift/scale/gate must be contiguous rows of length C")
    kind = {BF16: 0, F32: 1, torch.uint8: 2}.get(out.dtype)
    if kind is None:
        raise TypeError("layernorm: out must be bf16, fp32 or uint8 (e4m3)")
    if kind == 2 and (row_scale is None or row_scale.numel() != M):
        raise ValueError("layernorm: e4m3 output needs row_scale[M]")
    _call("ovd_resid_layernorm_modulate", _ptr(x, F32), ldx, _ptr(res, BF16), ldr, _ptr(res_gate, F32, True),
          _ptr(shift, F32, True), _ptr(scale, F32, True), _ptr(weight, BF16, True), _ptr(bias, BF16, True), _ptr(out),
          ldo, kind, _ptr(row_scale, F32, True), M, C, float(eps), int(round_norm), _stream())
    return out


def rmsnorm_rope(x, weight, out, head_dim, eps, rope=None, rows_per_sample=None, tok_offset=0, n_rope=0,
                 grid=(1, 1, 1)):
    M, C, ldi = _rows(x)
    Mo, Co, ldo = _rows(out)
    if (Mo, Co) != (M, C) or (weight is not None and weight.numel() != C):
        raise ValueError("rmsnorm: shape mismatch")
    rope_rows = 0
    if rope is not None:
        if rope.dtype != F32 or rope.dim() != 4 or rope.shape[0] != 3 or rope.shape[2] != head_dim // 2 \
                or rope.shape[3] != 2 or not rope.is_contiguous():
            raise ValueError("rmsnorm: rope table must be contiguous fp32 [3, rows, head_dim/2, 2]")
        rope_rows = rope.shape[1]
    _call("ovd_rmsnorm_rope_bf16", _ptr(x, BF16), ldi, _ptr(weight, BF16, True), _ptr(rope, F32, True), rope_rows,
          _ptr(out, BF16), ldo, M, C, head_dim, float(eps), rows_per_sample or max(M, 1), tok_offset, n_rope,
          grid[0], grid[1], grid[2], _stream())
    return out


def qkv_norm_rope(qkv, wq, wk, head_dim, eps, rope, rows_per_sample, tok_offset, n_rope, grid, send=None):
    """RMSNorm + RoPE of the q and k thirds of fused projection rows qkv [M, 3C] in one launch.
    send = None: in place.  send = [3, samples, p, rows, C/p] (contiguous): q, k (normalised, rotated) and v
    are written there split by destination rank - the send image of the sequence-parallel all-to-all."""
    M, C3, ld = _rows(qkv)
    C = C3 // 3
    if wq.numel() != C or wk.numel() != C:
        raise ValueError("qkv_norm_rope: weight length must be C")
    if rope.dtype != F32 or rope.dim() != 4 or rope.shape[0] != 3 or rope.shape[2] != head_dim // 2 \
            or rope.shape[3] != 2 or not rope.is_contiguous():
        raise ValueError("qkv_norm_rope: rope table must be contiguous fp32 [3, rows, head_dim/2, 2]")
    if send is None:
        out, ldo, part, cb, blk, samp, copy_v = qkv, ld, C, C, 0, rows_per_sample * ld, 0
    else:
        if send.dim() != 5 or not send.is_contiguous() or send.shape[0] != 3 or send.shape[3] != rows_per_sample \
                or send.shape[2] * send.shape[4] != C or send.shape[1] * rows_per_sample != M:
            raise ValueError("qkv_norm_rope: send buffer must be contiguous [3, samples, p, rows_per_sample, C/p]")
        _, ns, p, rows, cb = send.shape
        out, ldo, part, blk, samp, copy_v = send, cb, ns * p * rows * cb, rows * cb, p * rows * cb, 1
    _call("ovd_qkv_norm_rope_bf16", _ptr(qkv, BF16), ld, _ptr(wq, BF16), _ptr(wk, BF16), _ptr(rope, F32), rope.shape[1],
          _ptr(out, BF16), ldo, part, cb, blk, samp, copy_v, M, C, head_dim, float(eps), rows_per_sample, tok_offset,
          n_rope, grid[0], grid[1], grid[2], _stream())


def qkv_norm_rope_scatter(qkv, wq, wk, head_dim, eps, rope, rows_per_sample, tok_offset, n_rope, grid, block_ptrs,
                          total_rows, samples_total=None):
    """qkv_norm_rope whose column block j (C / p columns: the heads of rank j) goes to the buffer at
    block_ptrs[j] (int64 device tensor of p addresses): each a [3, samples, total_rows, C/p] bf16 receive
    buffer already advanced to this rank's first row.  The caller owns the validity of the addresses.
    samples_total: the receive buffers' sample count when the rows passed are only some of the samples (the
    addresses are then advanced to the first of them as well)."""
    M, C3, ld = _rows(qkv)
    C = C3 // 3
    p = block_ptrs.numel()
    if wq.numel() != C or wk.numel() != C or C % p or M % rows_per_sample:
        raise ValueError("qkv_norm_rope_scatter: shape mismatch")
    cb, ns = C // p, M // rows_per_sample
    if samples_total is not None:
        if samples_total < ns:
            raise ValueError("qkv_norm_rope_scatter: samples_total smaller than the samples passed")
        ns = int(samples_total)
    _call("ovd_qkv_norm_rope_scatter_bf16", _ptr(qkv, BF16), ld, _ptr(wq, BF16), _ptr(wk, BF16), _ptr(rope, F32),
          rope.shape[1], _ptr(block_ptrs, torch.int64), cb, ns * total_rows * cb, cb, total_rows * cb, M, C, head_dim,
          float(eps), rows_per_sample, tok_offset, n_rope, grid[0], grid[1], grid[2], _stream())


def gather_col_blocks_from(src_ptrs, dst, first=0):
    """dst [M, nblk * cb]: dst[m, b*cb + c] = (buffer at src_ptrs[b])[m, c]; src_ptrs an int64 device tensor of
    nblk addresses of contiguous bf16 [M, cb] blocks.  The caller owns the validity of the addresses."""
    nblk = src_ptrs.numel()
    M, Cd, ldd = _rows(dst)
    if Cd % nblk:
        raise ValueError("gather_col_blocks_from: the destination width does not split into the blocks")
    _call("ovd_gather_col_blocks_from_bf16", _ptr(src_ptrs, torch.int64), _ptr(dst, BF16), ldd, M, nblk, Cd // nblk,
          int(first) % nblk, _stream())
    return dst


def gather_col_blocks(src, dst):
    """src [nblk, M, cb] contiguous -> dst [M, nblk * cb] (row stride free): dst[m, b*cb + c] = src[b, m, c]."""
    nblk, M, cb = src.shape
    Md, Cd, ldd = _rows(dst)
    if not src.is_contiguous() or Md != M or Cd != nblk * cb:
        raise ValueError("gather_col_blocks: shape mismatch")
    _call("ovd_gather_col_blocks_bf16", _ptr(src, BF16), _ptr(dst, BF16), ldd, M, nblk, cb, _stream())
    return dst


def gemm(a, w, out, bias=None, epilogue=EPI_BF16, gate=None):
    M, K, lda = _rows(a)
    N, Kw, ldw = _rows(w)
    Mo, No, ldo = _rows(out)
    if Kw != K or Mo != M or No != N:
        raise ValueError(f"gemm: shape mismatch a[{M},{K}] w[{N},{Kw}] out[{Mo},{No}]")
    want = BF16 if epilogue in (EPI_BF16, EPI_BF16_GELU) else F32
    if bias is not None and (bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("gemm: bias must be a contiguous row of length N")
    if gate is not None and (gate.numel() != N or not gate.is_contiguous()):
        raise ValueError("gemm: gate must be a contiguous row of length N")
    _call("ovd_gemm_bf16", _ptr(a, BF16), lda, _ptr(w, BF16), ldw, _ptr(bias, BF16, True), _ptr(out, want), ldo,
          _ptr(gate, F32, True), M, N, K, epilogue, _stream())
    return out


def flash_attn_workspace_bytes(B, Lq, N, k_len):
    """Bytes of workspace with which flash_attn balances its last round of workgroups (0: no use)."""
    return int(lib().ovd_flash_attn_workspace_bytes(int(B), int(Lq), int(N), int(k_len)))


def flash_attn(q, k, v, out, k_len=None, softmax_scale=None, last_key_copies=1, workspace=None,
               static_workspace=False):
    """q [B,Lq,N,D], k/v [B,Lk,N,D], out [B,Lq,N,D] bf16; the (N,D) dims must be dense, the
    token and batch dims may be strided (views into a fused projection buffer).  Keys at or
    beyond k_len are masked; key k_len - 1 counts last_key_copies times (a run of identical
    trailing keys folded into one).  workspace: a device byte buffer of
    flash_attn_workspace_bytes(...) bytes (optional; see the header); static_workspace=True uses
    the library's built-in array instead (the allocation-free entry point)."""
    B, Lq, N, D = q.shape
    Lk = k.shape[1]
    for t in (q, k, v, out):
        if t.dim() != 4 or t.stride(3) != 1 or t.stride(2) != D:
            raise ValueError("attention: tensors must be [B,L,N,D] with dense (N,D)")
    if k.shape != v.shape or k.shape[0] != B or k.shape[2:] != q.shape[2:] or out.shape != q.shape:
        raise ValueError("attention: shape mismatch")
    scale = float(softmax_scale) if softmax_scale is not None else D ** -0.5
    kl = int(Lk if k_len is None else k_len)
    if static_workspace:
        if workspace is not None:
            raise ValueError("attention: pass either a workspace or static_workspace=True")
        _call("ovd_flash_attn_static_fwd_bf16", _ptr(q, BF16), _ptr(k, BF16), _ptr(v, BF16), _ptr(out, BF16), B, Lq,
              Lk, N, D, kl, int(last_key_copies), q.stride(1), k.stride(1), v.stride(1), out.stride(1), q.stride(0),
              k.stride(0), v.stride(0), out.stride(0), scale, _stream())
        return out
    if workspace is not None:
        if not workspace.is_cuda or not workspace.is_contiguous():
            raise ValueError("attention: workspace must be a contiguous device tensor")
        _call("ovd_flash_attn_ws_fwd_bf16", _ptr(q, BF16), _ptr(k, BF16), _ptr(v, BF16), _ptr(out, BF16), B, Lq, Lk,
              N, D, kl, int(last_key_copies), q.stride(1), k.stride(1), v.stride(1), out.stride(1), q.stride(0),
              k.stride(0), v.stride(0), out.stride(0), scale, workspace.data_ptr(),
              workspace.numel() * workspace.element_size(), _stream())
        return out
    if int(last_key_copies) != 1:
        _call("ovd_flash_attn_tail_fwd_bf16", _ptr(q, BF16), _ptr(k, BF16), _ptr(v, BF16), _ptr(out, BF16), B, Lq, Lk,
              N, D, kl, int(last_key_copies), q.stride(1), k.stride(1), v.stride(1), out.stride(1), q.stride(0),
              k.stride(0), v.stride(0), out.stride(0), scale, _stream())
        return out
    _call("ovd_flash_attn_fwd_bf16", _ptr(q, BF16), _ptr(k, BF16), _ptr(v, BF16), _ptr(out, BF16), B, Lq, Lk, N, D,
          kl, q.stride(1), k.stride(1), v.stride(1), out.stride(1),
          q.stride(0), k.stride(0), v.stride(0), out.stride(0), scale, _stream())
    return out


def quant_heads_i8(x, q, scale, rows_per_scale):
    """x bf16 [B, L, N, 128] (dense heads, strided tokens / batch) -> q int8 [B, L, N, 128], scale fp32
    [B, L, N] (rows_per_scale 1) or [B, ceil(L / 64), N] (rows_per_scale 64)."""
    B, L, N, D = x.shape
    for t in (x, q):
        if t.dim() != 4 or t.stride(3) != 1 or t.stride(2) != D:
            raise ValueError("quant_heads_i8: tensors must be [B,L,N,D] with dense (N,D)")
    if q.shape != x.shape or q.dtype != torch.int8:
        raise ValueError("quant_heads_i8: q must be int8 of x's shape")
    want = (B, L if rows_per_scale == 1 else -(-L // rows_per_scale), N)
    if tuple(scale.shape) != want or scale.dtype != F32 or not scale.is_contiguous():
        raise ValueError(f"quant_heads_i8: scale must be contiguous fp32 {want}")
    _call("ovd_quant_heads_i8", _ptr(x, BF16), x.stride(1), x.stride(0), _ptr(q, torch.int8), q.stride(1), q.stride(0),
          _ptr(scale, F32), B, L, N, D, int(rows_per_scale), _stream())


def flash_attn_qk8(q8, q_scale, k8, k_scale, v, out, k_len=None, softmax_scale=None, workspace=None):
    """Self-attention with Q K^T on int8 operands (quant_heads_i8 with rows_per_scale 1 for q, 64 for k)."""
    B, Lq, N, D = q8.shape
    Lk = k8.shape[1]
    for t in (q8, k8, v, out):
        if t.dim() != 4 or t.stride(3) != 1 or t.stride(2) != D:
            raise ValueError("attention: tensors must be [B,L,N,D] with dense (N,D)")
    if k8.shape != v.shape or out.shape != q8.shape or k8.shape[0] != B or k8.shape[2] != N:
        raise ValueError("attention: shape mismatch")
    if tuple(q_scale.shape) != (B, Lq, N) or tuple(k_scale.shape) != (B, -(-Lk // 64), N) \
            or not q_scale.is_contiguous() or not k_scale.is_contiguous():
        raise ValueError("attention: scales must be contiguous [B, Lq, N] and [B, ceil(Lk/64), N]")
    scale = float(softmax_scale) if softmax_scale is not None else D ** -0.5
    ws_ptr, ws_bytes = (None, 0) if workspace is None else (workspace.data_ptr(), workspace.numel() * workspace.element_size())
    _call("ovd_flash_attn_qk8_fwd_bf16", _ptr(q8, torch.int8), _ptr(q_scale, F32), _ptr(k8, torch.int8), _ptr(k_scale, F32),
          _ptr(v, BF16), _ptr(out, BF16), B, Lq, Lk, N, D, int(Lk if k_len is None else k_len), q8.stride(1), k8.stride(1),
          v.stride(1), out.stride(1), q8.stride(0), k8.stride(0), v.stride(0), out.stride(0), scale, ws_ptr, ws_bytes,
          _stream())
    return out


def patchify(lat, rows, patch):
    Cin, T, H, W = lat.shape
    if not lat.is_contiguous() or not rows.is_contiguous():
        raise ValueError("patchify: tensors must be contiguous")
    if lat.dtype not in (BF16, F32):
        raise TypeError("patchify: latents must be bf16 or fp32")
    pt, ph, pw = patch
    n = (T // pt) * (H // ph) * (W // pw)
    if tuple(rows.shape) != (n, Cin * pt * ph * pw):
        raise ValueError("patchify: rows buffer has the wrong shape")
    _call("ovd_patchify", _ptr(lat), int(lat.dtype == BF16), _ptr(rows, BF16), Cin, T, H, W, pt, ph, pw, _stream())
    return rows


def time_sinusoid(timesteps, out, step=0, step_ptr=None):
    _call("ovd_time_sinusoid", _ptr(timesteps, torch.float64), _ptr(step_ptr, torch.int32, True), int(step),
          _ptr(out, F32), out.numel(), _stream())
    return out


def gemv_f32(x, w, b, y, silu_in=False):
    N, K = w.shape
    if x.numel() != K or y.numel() != N or not (w.is_contiguous() and x.is_contiguous() and y.is_contiguous()):
        raise ValueError("gemv: shape mismatch")
    _call("ovd_gemv_f32", _ptr(x, F32), _ptr(w, BF16), _ptr(b, BF16, True), _ptr(y, F32), N, K, int(silu_in),
          _stream())
    return y


def modulation_table(param_ptrs, e0, out):
    n_layers, n6c = out.shape[0], out[0].numel()
    _call("ovd_modulation_table", _ptr(param_ptrs, torch.int64), _ptr(e0, F32), _ptr(out, F32), n_layers, n6c,
          _stream())
    return out


def head_modulation(mod, e, out):
    _call("ovd_head_modulation", _ptr(mod, BF16), _ptr(e, F32), _ptr(out, F32), e.numel(), _stream())
    return out


def head_unpatchify(y, w, bias, out, grid, patch, out_dim, tok_offset=0):
    """y rows 0.. hold tokens tok_offset..; writes those tokens' patches into out [Cout,T,H,W]."""
    M, C, ldy = _rows(y)
    n_tok = grid[0] * grid[1] * grid[2]
    rows = max(0, min(M, n_tok - tok_offset))
    if not out.is_contiguous():
        raise ValueError("head: out must be contiguous")
    _call("ovd_head_unpatchify", _ptr(y, F32), ldy, _ptr(w, BF16), _ptr(bias, BF16, True), _ptr(out, F32), rows, C,
          out_dim, patch[0], patch[1], patch[2], grid[0], grid[1], grid[2], tok_offset, _stream())
    return out


def cfg_unipc_step(cond, uncond, sample, last_sample, m1, m2, coef, guide_scale, step=0, step_ptr=None):
    n = sample.numel()
    for t in (cond, uncond, m1, m2):
        if t.numel() != n or not t.is_contiguous():
            raise ValueError("cfg_unipc: buffers must be contiguous and equally sized")
    if last_sample.dtype != sample.dtype or sample.dtype not in (BF16, F32):
        raise TypeError("cfg_unipc: sample/last_sample must share a bf16 or fp32 dtype")
    _call("ovd_cfg_unipc_step", _ptr(cond, F32), _ptr(uncond, F32), _ptr(sample), _ptr(last_sample), _ptr(m1, F32),
          _ptr(m2, F32), _ptr(coef, F32), float(guide_scale), _ptr(step_ptr, torch.int32, True), int(step), n,
          int(sample.dtype == BF16), _stream())


def cfg_dpm_step(cond, uncond, sample, nxt, m1, coef, guide_scale, step=0, step_ptr=None):
    """Fused guidance + DPM-Solver++ step: reads `sample` (bf16 or fp32), writes the fp32 `nxt`
    (which may be `sample` itself when that is fp32) and the x0 history `m1`."""
    n = sample.numel()
    for t in (cond, uncond, m1, nxt, sample):
        if t.numel() != n or not t.is_contiguous():
            raise ValueError("cfg_dpm: buffers must be contiguous and equally sized")
    if sample.dtype not in (BF16, F32):
        raise TypeError("cfg_dpm: sample must be bf16 or fp32")
    _call("ovd_cfg_dpm_step", _ptr(cond, F32), _ptr(uncond, F32), _ptr(sample), _ptr(nxt, F32), _ptr(m1, F32),
          _ptr(coef, F32), float(guide_scale), _ptr(step_ptr, torch.int32, True), int(step), n,
          int(sample.dtype == BF16), _stream())


def t5_attention(qkv, pos_embedding, bucket, k_lens, out, num_heads):
    """qkv [B, L, 3*N*64] bf16, pos_embedding [buckets, N] bf16, bucket [L, L] int32, k_lens [B] int32
    or None, out [B, L, N*64] bf16."""
    B, L, W = qkv.shape
    if W != 3 * num_heads * 64 or qkv.stride(2) != 1 or qkv.stride(0) != L * qkv.stride(1):
        raise ValueError("t5 attention: qkv must be [B, L, 3*N*64] with dense rows")
    if tuple(out.shape) != (B, L, num_heads * 64) or not out.is_contiguous():
        raise ValueError("t5 attention: out must be contiguous [B, L, N*64]")
    if tuple(bucket.shape) != (L, L) or bucket.dtype != torch.int32 or not bucket.is_contiguous():
        raise ValueError("t5 attention: bucket must be contiguous int32 [L, L]")
    if pos_embedding.shape[1] != num_heads or not pos_embedding.is_contiguous():
        raise ValueError("t5 attention: pos_embedding must be contiguous [buckets, N]")
    _call("ovd_t5_attention_bf16", _ptr(qkv, BF16), qkv.stride(1), _ptr(pos_embedding, BF16), _ptr(bucket, torch.int32),
          _ptr(k_lens, torch.int32, True), _ptr(out, BF16), out.stride(1), B, L, num_heads, 64, _stream())
    return out


def t5_gated_gelu(h, out):
    M, F2, ldh = _rows(h)
    Mo, F, ldo = _rows(out)
    if Mo != M or F2 != 2 * F:
        raise ValueError("t5 gated gelu: h must be [M, 2F] and out [M, F]")
    _call("ovd_t5_gated_gelu_bf16", _ptr(h, BF16), ldh, _ptr(out, BF16), ldo, M, F, _stream())
    return out


def add_bf16(x, y):
    if x.shape != y.shape or not x.is_contiguous() or not y.is_contiguous():
        raise ValueError("add_bf16: tensors must be contiguous and equally shaped")
    _call("ovd_add_bf16", _ptr(x, BF16), _ptr(y, BF16), x.numel(), _stream())
    return x


def step_advance(step_ptr):
    _call("ovd_step_advance", _ptr(step_ptr, torch.int32), _stream())


# ---------------------------------------------------------------------------- VAE (channels-last fp32)

def _f32c(t, allow_none=False):
    if t is None:
        return _ptr(None, allow_none=allow_none)
    if not t.is_contiguous():
        raise ValueError("this operator needs contiguous tensors")
    return _ptr(t, F32)


class SplitWeight:
    """Convolution weights [taps, Cout, Cin] in the split-bf16 form the conv kernel keeps in LDS
    (ovd_vae_split_weights): `data` is uint16 [taps * Cout, ceil(Cin / 32), 64]."""

    def __init__(self, w):
        if w.dim() != 3 or w.dtype != F32 or not w.is_contiguous():
            raise ValueError("SplitWeight: expected contiguous fp32 [taps, Cout, Cin]")
        self.shape = tuple(w.shape)
        taps, cout, cin = self.shape
        self.data = torch.empty(taps * cout, (cin + 31) // 32, 64, dtype=torch.int16, device=w.device)
        _call("ovd_vae_split_weights", _f32c(w), self.data.data_ptr(), taps * cout, cin, _stream())


def vae_conv3d(x, w, bias, kernel, prev2=None, prev1=None, rnorm=None, rnorm_prev2=None, rnorm_prev1=None,
               gamma=None, silu=False, resid=None, stride_t=1, stride_hw=1, pad_t=0, pad_hw=0, upsample2=False,
               time_interleave=False, out=None):
    """x [T,H,W,Cin] fp32, w [KT*KH*KW, Cout, Cin] fp32 -> [To,Ho,Wo,Cout] (or the frame-interleaved
    [2To,Ho,Wo,Cout/2]).  prev1/prev2 are the [H,W,Cin] frames at time -1/-2 (None = zeros).
    w may be a SplitWeight (same results, cheaper weight staging)."""
    T, H, W, Cin = x.shape
    KT, KH, KW = kernel
    taps, Cout, Cin_w = w.shape
    if taps != KT * KH * KW or Cin_w != Cin:
        raise ValueError(f"conv3d: weight {tuple(w.shape)} does not match kernel {kernel} / Cin {Cin}")
    for p in (prev1, prev2):
        if p is not None and tuple(p.shape) != (H, W, Cin):
            raise ValueError("conv3d: previous-chunk frame has the wrong shape")
    Hin, Win = (2 * H, 2 * W) if upsample2 else (H, W)
    Ho = (Hin + 2 * pad_hw - KH + (1 if stride_hw == 2 else 0)) // stride_hw + 1
    Wo = (Win + 2 * pad_hw - KW + (1 if stride_hw == 2 else 0)) // stride_hw + 1
    To = (T + pad_t - KT) // stride_t + 1
    shape = (2 * To, Ho, Wo, Cout // 2) if time_interleave else (To, Ho, Wo, Cout)
    if out is None:
        out = torch.empty(shape, dtype=F32, device=x.device)
    elif tuple(out.shape) != shape:
        raise ValueError(f"conv3d: out has shape {tuple(out.shape)}, expected {shape}")
    if resid is not None and tuple(resid.shape) != shape:
        raise ValueError("conv3d: residual shape mismatch")
    split = isinstance(w, SplitWeight)
    _call("ovd_vae_conv3d_sw_f32" if split else "ovd_vae_conv3d_f32", _f32c(x), _f32c(prev2, True),
          _f32c(prev1, True), _f32c(rnorm, True),
          _f32c(rnorm_prev2, True), _f32c(rnorm_prev1, True), _f32c(gamma, True), int(silu),
          w.data.data_ptr() if split else _f32c(w),
          _f32c(bias, True), _f32c(resid, True), _f32c(out), T, H, W, Cin, Cout, KT, KH, KW, stride_t, stride_hw,
          pad_t, pad_hw, int(upsample2), int(time_interleave), _stream())
    return out


def vae_act_split(x, gamma=None, silu=False):
    """x fp32 [T,H,W,C] (C % 32 == 0) -> int16 [T,H,W,C/32,64]: SiLU(RMS_norm(x)) (each optional) as the
    split-bf16 image the convolution kernel stages (ovd_vae_act_split)."""
    C = x.shape[-1]
    if C % 32:
        raise ValueError("vae_act_split: C must be a multiple of 32")
    out = torch.empty(tuple(x.shape[:-1]) + (C // 32, 64), dtype=torch.int16, device=x.device)
    _call("ovd_vae_act_split", _f32c(x), _f32c(gamma, True), int(silu), out.data_ptr(), x.numel() // C, C, _stream())
    return out


def vae_conv3d_ps(xs, w, bias, kernel, prev2=None, prev1=None, resid=None, pad_t=0, pad_hw=0, upsample2=False, out=None):
    """The 3x3 stride-1 convolution on a pre-activated, pre-split input (vae_act_split): xs int16
    [T,H,W,Cin/32,64], w a SplitWeight [KT*9, Cout, Cin]; prev1 / prev2 the [H,W,Cin/32,64] frames at time -1 / -2;
    upsample2: convolve the nearest-neighbour 2x spatial upsampling of the image."""
    T, H, W, nch, _ = xs.shape
    KT, KH, KW = kernel
    taps, Cout, Cin = w.shape
    if not isinstance(w, SplitWeight) or taps != KT * KH * KW or Cin != nch * 32 or xs.dtype != torch.int16 \
            or not xs.is_contiguous():
        raise ValueError("conv3d_ps: needs a contiguous int16 split input and matching SplitWeight weights")
    for p in (prev1, prev2):
        if p is not None and (tuple(p.shape) != (H, W, nch, 64) or p.dtype != torch.int16 or not p.is_contiguous()):
            raise ValueError("conv3d_ps: previous-chunk frame has the wrong shape")
    Hin, Win = (2 * H, 2 * W) if upsample2 else (H, W)
    shape = (T + pad_t - KT + 1, Hin + 2 * pad_hw - KH + 1, Win + 2 * pad_hw - KW + 1, Cout)
    if out is None:
        out = torch.empty(shape, dtype=F32, device=xs.device)
    elif tuple(out.shape) != shape:
        raise ValueError(f"conv3d_ps: out has shape {tuple(out.shape)}, expected {shape}")
    if resid is not None and tuple(resid.shape) != shape:
        raise ValueError("conv3d_ps: residual shape mismatch")
    _call("ovd_vae_conv3d_ps_f32", xs.data_ptr(), None if prev2 is None else prev2.data_ptr(),
          None if prev1 is None else prev1.data_ptr(), w.data.data_ptr(), _f32c(bias, True), _f32c(resid, True), _f32c(out),
          T, H, W, Cin, Cout, KT, KH, KW, pad_t, pad_hw, int(upsample2), _stream())
    return out


def vae_voxel_rnorm(x):
    C = x.shape[-1]
    out = torch.empty(x.shape[:-1], dtype=F32, device=x.device)
    _call("ovd_vae_voxel_rnorm", _f32c(x), _f32c(out), x.numel() // C, C, _stream())
    return out


def vae_to_channels_last(x, c_pad=None, pre=None, mul=None, post=None):
    """[C,T,H,W] -> [T,H,W,c_pad] fp32."""
    C, T, H, W = x.shape
    cp = C if c_pad is None else c_pad
    out = torch.empty(T, H, W, cp, dtype=F32, device=x.device)
    _call("ovd_vae_to_channels_last", _f32c(x), _f32c(out), _f32c(pre, True), _f32c(mul, True), _f32c(post, True), C,
          cp, T * H * W, _stream())
    return out


def vae_from_channels_last(x, channels=None, pre=None, mul=None, post=None, clamp=False):
    """[T,H,W,Cs] -> [C,T,H,W] (first C channels)."""
    T, H, W, Cs = x.shape
    C = Cs if channels is None else channels
    out = torch.empty(C, T, H, W, dtype=F32, device=x.device)
    _call("ovd_vae_from_channels_last", _f32c(x), _f32c(out), _f32c(pre, True), _f32c(mul, True), _f32c(post, True),
          C, Cs, T * H * W, int(clamp), _stream())
    return out


def video_to_uint8(x, channels_last, value_range=(-1.0, 1.0), channels=3):
    """fp32 video -> uint8 [T,H,W,C]; x is channels-last [T,H,W,Cs] or planar [C,T,H,W]."""
    if channels_last:
        T, Hh, W, Cs = x.shape
        vs, cs = Cs, 1
    else:
        C_, T, Hh, W = x.shape
        if C_ != channels:
            raise ValueError("video_to_uint8: planar input must have exactly `channels` channels")
        vs, cs = 1, T * Hh * W
    out = torch.empty(T, Hh, W, channels, dtype=torch.uint8, device=x.device)
    _call("ovd_video_to_uint8", _f32c(x), out.data_ptr(), vs, cs, channels, T * Hh * W, float(value_range[0]),
          float(value_range[1]), _stream())
    return out


EPI_FP8_GELU = 4


def _u8(t):
    if t.dtype != torch.uint8 or not t.is_cuda or t.stride(-1) != 1:
        raise ValueError("expected a uint8 (e4m3 bit pattern) tensor on the HIP device")
    return t.data_ptr()


def quant_rows_fp8(x, q=None, scale=None):
    """bf16 [M, K] -> (e4m3 bytes [M, K] uint8, fp32 scale [M]): x ~ q * scale[:, None]."""
    M, K = x.shape
    if q is None:
        q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty(M, dtype=F32, device=x.device)
    _call("ovd_quant_rows_fp8", _ptr(x, BF16), x.stride(0), _u8(q), q.stride(0), _f32c(scale), M, K, _stream())
    return q, scale


def layernorm_modulate_fp8(x, out, row_scale, eps, shift=None, scale=None, weight=None, bias=None, round_norm=False):
    M, C = x.shape
    _call("ovd_layernorm_modulate_fp8", _f32c(x), x.stride(0), _f32c(shift, True), _f32c(scale, True),
          _ptr(weight, BF16, True), _ptr(bias, BF16, True), _u8(out), out.stride(0), _f32c(row_scale), M, C,
          float(eps), int(round_norm), _stream())
    return out


def gemm_fp8(a8, w8, w_scale, out, bias=None, epilogue=EPI_BF16, a_scale=None, a_scale_all=1.0, gate=None,
             out_scale=1.0):
    """out = epilogue((a8 @ w8.T) * a_scale[:, None] * a_scale_all * w_scale[None, :] + bias)."""
    M, K = a8.shape
    N = w8.shape[0]
    if w8.shape[1] != K or out.shape[0] != M or out.shape[1] != N:
        raise ValueError("gemm_fp8: shape mismatch")
    want = {EPI_BF16: BF16, EPI_BF16_GELU: BF16, EPI_F32: F32, EPI_RESID: F32, EPI_FP8_GELU: torch.uint8}[epilogue]
    if out.dtype != want or out.stride(1) != 1:
        raise ValueError("gemm_fp8: output dtype/stride does not match the epilogue")
    _call("ovd_gemm_fp8", _u8(a8), a8.stride(0), _f32c(a_scale, True), float(a_scale_all), _u8(w8), w8.stride(0),
          _f32c(w_scale), _ptr(bias, BF16, True), out.data_ptr(), out.stride(0), _f32c(gate, True), float(out_scale),
          M, N, K, epilogue, _stream())
    return out


def resize_coeffs(in_size, out_size, device):
    """Device tap tables (bounds int32 [out, 2], coef int32 [out, ksize]) of one resize axis."""
    L = lib()
    ksize = L.ovd_resize_coeffs(in_size, out_size, None, None, 0)
    if ksize <= 0:
        raise ValueError(f"resize {in_size} -> {out_size} is not served (empty, or more than 31x reduction)")
    bounds = torch.empty(out_size, 2, dtype=torch.int32)
    coef = torch.empty(out_size, ksize, dtype=torch.int32)
    if L.ovd_resize_coeffs(in_size, out_size, bounds.data_ptr(), coef.data_ptr(), ksize) != ksize:
        raise RuntimeError("ovd_resize_coeffs: inconsistent tap-table width")
    return bounds.to(device), coef.to(device), ksize


def frames_resize_normalize(frames, crop, out_hw, lut):
    """frames uint8 [T,H,W,3] on the device -> fp32 [3,T,out_h,out_w]; crop = (top, left, h, w)."""
    if frames.dtype != torch.uint8 or not frames.is_cuda or not frames.is_contiguous() or frames.dim() != 4 \
            or frames.shape[3] != 3:
        raise ValueError("frames must be a contiguous uint8 [T,H,W,3] tensor on the HIP device")
    T, Hh, W, _ = frames.shape
    top, left, ch, cw = crop
    oh, ow = out_hw
    if lut.numel() != 256:
        raise ValueError("lut must hold 256 floats")
    bw, kw, ksw = resize_coeffs(cw, ow, frames.device)
    bh, kh, ksh = resize_coeffs(ch, oh, frames.device)
    tmp = torch.empty(T, ch, ow, 3, dtype=torch.uint8, device=frames.device)
    out = torch.empty(3, T, oh, ow, dtype=F32, device=frames.device)
    _call("ovd_frames_resize_normalize", frames.data_ptr(), tmp.data_ptr(), out.data_ptr(), bw.data_ptr(),
          kw.data_ptr(), ksw, bh.data_ptr(), kh.data_ptr(), ksh, _f32c(lut), T, Hh, W, top, left, ch, cw, oh, ow,
          _stream())
    return out


def softmax_rows(s, scale):
    M, N = s.shape
    _call("ovd_softmax_rows_f32", _f32c(s), M, N, float(scale), _stream())
    return s


# ---------------------------------------------------------------------------- peer-mapped exchange
# (set-up calls take and return raw device addresses as Python ints; distributed/peer_mapped.py owns them)

def peer_alloc(nbytes, fine_grained=False):
    out = ctypes.c_void_p()
    _call("ovd_peer_alloc", ctypes.byref(out), int(nbytes), 1 if fine_grained else 0)
    return int(out.value)


def peer_free(addr):
    _call("ovd_peer_free", ctypes.c_void_p(addr))


def peer_export(addr):
    buf = (ctypes.c_ubyte * 64)()
    _call("ovd_peer_export", ctypes.c_void_p(addr), buf)
    return bytes(buf)


def peer_open(handle):
    if len(handle) != 64:
        raise ValueError("an exported buffer handle is 64 bytes")
    out = ctypes.c_void_p()
    _call("ovd_peer_open", (ctypes.c_ubyte * 64).from_buffer_copy(handle), ctypes.byref(out))
    return int(out.value)


def peer_close(addr):
    _call("ovd_peer_close", ctypes.c_void_p(addr))


def peer_barrier(my_flags_addr, peer_flags_table, state, p, rank, timeout_s):
    _call("ovd_peer_barrier", ctypes.c_void_p(my_flags_addr), _ptr(peer_flags_table, torch.int64),
          _ptr(state, torch.int32), p, rank, float(timeout_s), _stream())


def peer_all_to_all(recv, peer_send_table, chunk_bytes, p, rank):
    if not recv.is_contiguous() or recv.numel() * recv.element_size() != p * chunk_bytes:
        raise ValueError("peer_all_to_all: recv must be contiguous and hold p chunks")
    _call("ovd_peer_all_to_all", _ptr(recv), _ptr(peer_send_table, torch.int64), int(chunk_bytes), p, rank, _stream())


def peer_sum_f32(out, peer_src_table, p):
    if not out.is_contiguous():
        raise ValueError("peer_sum: out must be contiguous")
    _call("ovd_peer_sum_f32", _ptr(out, F32), _ptr(peer_src_table, torch.int64), out.numel(), p, _stream())
