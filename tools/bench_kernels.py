#!/usr/bin/env python
"""Micro-benchmarks of the two kernels that carry the FLOPs (attention, GEMM) at the
cfg2 shapes, timed with HIP events on torch's current stream.  Development aid; the judged
numbers come from bench.py."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omni-video_amd"))
import torch                                   # noqa: E402
from omnivideo import _hip as H                # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, iters=10, warm=2):
    # the first launches after an idle period run at a low clock: warm up for at least 0.3 s
    import time
    t0 = time.time()
    n = 0
    while n < warm or time.time() - t0 < 0.3:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def attn(B, Lq, Lk, N, k_len=None, check=False, strided_q=False, fused_kv=False, same_rows=0, balance=False):
    D = 128
    if strided_q:      # q addressed inside a fused [B, Lq, 3*N*D] projection buffer
        q = torch.randn(B, Lq, 3 * N * D, device="cuda").to(BF16)[:, :, :N * D].unflatten(2, (N, D))
    else:
        q = torch.randn(B, Lq, N, D, device="cuda").to(BF16)
    k = torch.randn(B, Lk, N, D, device="cuda").to(BF16)
    v = torch.randn(B, Lk, N, D, device="cuda").to(BF16)
    if fused_kv:       # k | v halves of one [B, Lk, 2*N*D] buffer (the cached cross-attention K/V)
        kv = torch.randn(B, Lk, 2 * N * D, device="cuda").to(BF16)
        if same_rows:
            kv[:, same_rows:] = kv[:, same_rows:same_rows + 1]
        k, v = kv[:, :, :N * D].unflatten(2, (N, D)), kv[:, :, N * D:].unflatten(2, (N, D))
    o = torch.empty(B, Lq, N, D, device="cuda", dtype=BF16)
    ws = None
    if balance:
        nbytes = H.flash_attn_workspace_bytes(B, Lq, N, k_len or Lk)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda") if nbytes else None
    ms = timeit(lambda: H.flash_attn(q, k, v, o, k_len=k_len, workspace=ws))
    fl = 4.0 * B * N * Lq * (k_len or Lk) * D
    msg = f"attn B{B} Lq{Lq} Lk{Lk} N{N}{' balanced' if ws is not None else ''}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s"
    if check:
        ref = torch.nn.functional.scaled_dot_product_attention(
            q.transpose(1, 2).float(), k.transpose(1, 2).float()[:, :, :k_len], v.transpose(1, 2).float()[:, :, :k_len])
        err = (o.float() - ref.transpose(1, 2)).abs().max().item()
        msg += f"  max|err| vs fp32 SDPA {err:.2e}"
    print(msg, flush=True)


def attn_qk8(B, L, N):
    """self-attention with int8 Q K^T beside the bf16 kernel on the same inputs (both with tail balancing)"""
    D = 128
    q, k, v = (torch.randn(B, L, N, D, device="cuda").to(BF16) for _ in range(3))
    q8, k8 = (torch.empty(B, L, N, D, device="cuda", dtype=torch.int8) for _ in range(2))
    qs = torch.empty(B, L, N, device="cuda")
    ks = torch.empty(B, -(-L // 64), N, device="cuda")
    o = torch.empty_like(q)
    nbytes = H.flash_attn_workspace_bytes(B, L, N, L)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda") if nbytes else None
    fl = 4.0 * B * N * L * L * D
    ms = timeit(lambda: H.flash_attn(q, k, v, o, workspace=ws))
    def quant():
        H.quant_heads_i8(q, q8, qs, 1)
        H.quant_heads_i8(k, k8, ks, 64)
    msq = timeit(quant)
    quant()
    ms8 = timeit(lambda: H.flash_attn_qk8(q8, qs, k8, ks, v, o, workspace=ws))
    print(f"attn B{B} L{L} N{N}: bf16 {ms:.3f} ms {fl / ms / 1e9:.0f} TFLOP/s | int8 QK^T {ms8:.3f} ms "
          f"{fl / ms8 / 1e9:.0f} TFLOP/s (+ quantise q,k {msq * 1e3:.0f} us) | ratio {ms / (ms8 + msq):.3f}", flush=True)


def gemm(M, N, K, epi=0, check=False, uniform=False):
    if uniform:     # operands uniform in [-1, 1): the distribution published GEMM numbers are quoted on
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(BF16)
        w = (torch.rand(N, K, device="cuda") * 2 - 1).to(BF16)
    else:
        a = torch.randn(M, K, device="cuda").to(BF16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(BF16)
    b = torch.randn(N, device="cuda").to(BF16)
    out = torch.empty(M, N, device="cuda", dtype=BF16 if epi < 2 else torch.float32)
    if epi >= 2:
        out.zero_()
    ms = timeit(lambda: H.gemm(a, w, out, b, epi))
    msg = f"gemm M{M} N{N} K{K} epi{epi}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s"
    if check:
        if epi >= 2:
            out.zero_()
        H.gemm(a, w, out, b, epi)
        ref = (a.float() @ w.float().t() + b.float())
        if epi == 1:
            ref = torch.nn.functional.gelu(ref.to(BF16).float(), approximate="tanh")
        err = (out.float() - ref.to(BF16).float()).abs().max().item()
        msg += f"  max|err| vs torch {err:.3e} (ref max {ref.abs().max().item():.2f})"
    print(msg, flush=True)


def gemm8(M, N, K, epi=0, check=False):
    """e4m3 GEMM: operands quantised per row from the same distributions as gemm()."""
    a = torch.randn(M, K, device="cuda").to(BF16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(BF16)
    a8, sa = H.quant_rows_fp8(a)
    w8, sw = H.quant_rows_fp8(w)
    b = torch.randn(N, device="cuda").to(BF16)
    dt = {0: BF16, 1: BF16, 2: torch.float32, 3: torch.float32, 4: torch.uint8}[epi]
    out = torch.zeros(M, N, device="cuda", dtype=dt)
    ms = timeit(lambda: H.gemm_fp8(a8, w8, sw, out, b, epi, a_scale=sa))
    msg = f"gemm_fp8 M{M} N{N} K{K} epi{epi}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s"
    if check:
        out.zero_()
        H.gemm_fp8(a8, w8, sw, out, b, epi, a_scale=sa)
        f8 = torch.float8_e4m3fn
        ref = (a8.view(f8).float() * sa[:, None]) @ (w8.view(f8).float() * sw[:, None]).t() + b.float()
        full = a.float() @ w.float().t() + b.float()
        err = (out.float() - ref.to(BF16).float()).abs().max().item()
        rel = ((out.float() - full).norm() / full.norm()).item()
        msg += f"  max|err| vs dequantised product {err:.3e}; rel L2 vs bf16-operand product {rel:.3e}"
    print(msg, flush=True)


def vae(latent=(16, 13, 60, 104), encode=False):
    """Full-width (dim 96) VAE decode (and encode) of one cfg2-sized latent, random weights."""
    import time
    from omnivideo.modules.vae2_1 import Wan2_1_VAE
    with torch.device("cuda"):
        v = Wan2_1_VAE(z_dim=16, vae_pth=None, device="cuda")
    z = torch.randn(*latent, device="cuda")
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        x = v.decode([z])[0]
        torch.cuda.synchronize()
        print(f"vae decode {latent} -> {tuple(x.shape)}: {time.time() - t0:.3f} s", flush=True)
    if encode:
        for it in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            z2 = v.encode([x])[0]
            torch.cuda.synchronize()
            print(f"vae encode {tuple(x.shape)} -> {tuple(z2.shape)}: {time.time() - t0:.3f} s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["attn", "gemm"])
    args = ap.parse_args()
    if "vae" in args.what:
        vae(encode=True)
    if "vae_small" in args.what:
        vae((16, 3, 30, 52), encode=True)
    if "attn" in args.what:
        attn(1, 1024, 1024, 2, 1000, check=True)
        attn(2, 20280, 20280, 12)
        attn(2, 20280, 20280, 12, balance=True)
        attn(2, 20280, 6144, 12)
    if "attn8" in args.what:
        attn_qk8(2, 20280, 12)
        attn_qk8(2, 75600, 5)
    if "attnbal" in args.what:      # one self-attention shape alone (for counter passes); default: cfg2's
        B_, L_, N_ = (int(v) for v in os.environ.get("OVD_BENCH_ATTN_SHAPE", "2,20280,12").split(","))
        attn(B_, L_, L_, N_, balance=True)
    if "xattn131" in args.what:     # cross-attention as the cfg2 model runs it: 131 keys read, fused K|V cache
        for kl in (131, 512, 1024, 2048):
            attn(2, 20280, 6144, 12, k_len=kl, strided_q=True, fused_kv=True)
    if "xattn" in args.what:
        attn(2, 20280, 512, 12)
        attn(2, 20280, 512, 12, strided_q=True)
        attn(2, 20280, 64, 12, strided_q=True)
        attn(2, 20280, 512, 12, strided_q=True, fused_kv=True)
        attn(2, 20280, 512, 12, strided_q=True, fused_kv=True, same_rows=20)
    if "gemmcheck" in args.what:
        gemm(1000, 512, 128, 0, check=True)
        gemm(777, 256, 192, 3, check=True)
        gemm(4096, 1536, 1536, 1, check=True)
        gemm(40560, 1536, 1536, 2, check=True)
    if "gemmshapes" in args.what:   # the five Linear shapes of the cfg2 model on the library as loaded
        M = 40560
        line = []
        for (N, K, epi) in ((4608, 1536, 0), (1536, 1536, 0), (8960, 1536, 1), (1536, 8960, 0), (1536, 8960, 3)):
            a = torch.randn(M, K, device="cuda").to(BF16)
            w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(BF16)
            b = torch.randn(N, device="cuda").to(BF16)
            out = torch.zeros(M, N, device="cuda", dtype=BF16 if epi < 2 else torch.float32)
            t = min(timeit(lambda: H.gemm(a, w, out, b, epi), iters=20) for _ in range(3))
            line.append(f"N{N}K{K}e{epi} {2.0 * M * N * K / 1e9 / t:.0f}")
            del a, w, out
        print(os.environ.get("OVD_LIB_PATH", "default").split("/")[-1], " | ".join(line), flush=True)
    if "gemmsp" in args.what:       # the Linear shapes of a sequence-parallel rank of the 1.3B model (2 and 4 ranks)
        for M in (20280, 10140):
            line = []
            for (N, K, epi) in ((4608, 1536, 0), (1536, 1536, 0), (8960, 1536, 1), (1536, 8960, 0)):
                a = torch.randn(M, K, device="cuda").to(BF16)
                w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(BF16)
                b = torch.randn(N, device="cuda").to(BF16)
                out = torch.zeros(M, N, device="cuda", dtype=BF16)
                t = min(timeit(lambda: H.gemm(a, w, out, b, epi), iters=20) for _ in range(3))
                line.append(f"N{N}K{K}e{epi} {2.0 * M * N * K / 1e9 / t:.0f}")
            print(f"M{M} min_tiles={os.environ.get('OVD_GEMM256_MIN_TILES', 'default')}", " | ".join(line), flush=True)
    if "twostream" in args.what:    # does a second stream's kernel fill the partly empty last round of tiles?
        C, F, L = 1536, 8960, 20280
        def layer_ops(M, tag):
            x = torch.randn(M, C, device="cuda").to(BF16)
            wq = (torch.randn(3 * C, C, device="cuda") / 40).to(BF16)
            wo = (torch.randn(C, C, device="cuda") / 40).to(BF16)
            w1 = (torch.randn(F, C, device="cuda") / 40).to(BF16)
            w2 = (torch.randn(C, F, device="cuda") / 90).to(BF16)
            qkv = torch.empty(M, 3 * C, device="cuda", dtype=BF16)
            o = torch.empty(M, C, device="cuda", dtype=BF16)
            h = torch.empty(M, F, device="cuda", dtype=BF16)
            B = M // L
            att = torch.empty(B, L, 12, 128, device="cuda", dtype=BF16)
            def run():
                H.gemm(x, wq, qkv, None, 0)
                q5 = qkv.view(B, L, 3, 12, 128)
                H.flash_attn(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], att)
                H.gemm(att.view(M, C), wo, o, None, 0)
                H.gemm(o, wo, x, None, 0)
                H.gemm(x, wo, o, None, 0)
                H.gemm(o, w1, h, None, 1)
                H.gemm(h, w2, x, None, 0)
            return run
        both = layer_ops(2 * L, "both")
        a, b = layer_ops(L, "a"), layer_ops(L, "b")
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        def two():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                a()
            with torch.cuda.stream(s2):
                b()
            cur.wait_stream(s1); cur.wait_stream(s2)
        def seq():
            a(); b()
        for name, fn in (("one stream, both branches batched", both), ("one stream, branch after branch", seq),
                         ("two streams, one branch each", two)):
            g = torch.cuda.CUDAGraph()
            fn(); torch.cuda.synchronize()
            with torch.cuda.graph(g):
                fn()
            ms = min(timeit(g.replay, iters=10) for _ in range(3))
            print(f"layer GEMMs + self-attention, {name}: {ms:.3f} ms", flush=True)
    if "gemmsq" in args.what:
        gemm(4096, 4096, 4096, 0, uniform=True)
        gemm(8192, 8192, 8192, 0, uniform=True)
        gemm(8192, 8192, 8192, 0)
    if "gemm8check" in args.what:
        gemm8(1000, 512, 128, 0, check=True)
        gemm8(777, 256, 192, 3, check=True)
        gemm8(4096, 1536, 1536, 0, check=True)
    if "gemm8" in args.what:
        M = 40560
        gemm8(M, 4608, 1536, 0)
        gemm8(M, 1536, 1536, 3)
        gemm8(M, 8960, 1536, 4)
        gemm8(M, 1536, 8960, 3)
        gemm8(8192, 8192, 8192, 0)
    if "gemmepi" in args.what:      # what each epilogue costs on the K = 1536 shapes
        for epi in (0, 1, 2, 3):
            gemm(40560, 1536, 1536, epi)
        gemm(40560, 4608, 1536, 0)
        gemm(40560, 4608, 1536, 2)
    if "gemmk" in args.what:        # fixed cost per output tile: the same N, M at growing K
        for K in (128, 256, 512, 1536):
            gemm(40560, 8960, K, 1)
        for K in (128, 256, 512, 1536):
            gemm(40560, 8960, K, 0)
    if "lnres" in args.what:        # LayerNorm-modulate with and without the fused residual update
        for M, C in ((20480, 1536), (40560, 1536), (75600, 5120)):
            x = torch.randn(M, C, device="cuda")
            y = torch.randn(M, C, device="cuda").to(BF16)
            sh, sc, gt = (torch.randn(C, device="cuda") * 0.1 for _ in range(3))
            w, b = (torch.randn(C, device="cuda").to(BF16) for _ in range(2))
            h = torch.empty(M, C, device="cuda", dtype=BF16)
            ms = timeit(lambda: H.layernorm_modulate(x, h, 1e-6, shift=sh, scale=sc))
            print(f"ln_mod {M}x{C}: {ms * 1e3:.1f} us  {M * C * 6 / ms / 1e9:.2f} TB/s")
            ms = timeit(lambda: H.resid_layernorm_modulate(x, y, h, 1e-6, res_gate=gt, shift=sh, scale=sc))
            print(f"resid ln_mod {M}x{C}: {ms * 1e3:.1f} us  {M * C * 12 / ms / 1e9:.2f} TB/s")
            ms = timeit(lambda: H.resid_layernorm_modulate(x, y, h, 1e-6, res_gate=gt, weight=w, bias=b))
            print(f"resid ln affine {M}x{C}: {ms * 1e3:.1f} us  {M * C * 12 / ms / 1e9:.2f} TB/s")
            h8 = torch.empty(M, C, device="cuda", dtype=torch.uint8)
            rs = torch.empty(M, device="cuda")
            ms = timeit(lambda: H.resid_layernorm_modulate(x, y, h8, 1e-6, res_gate=gt, shift=sh, scale=sc, row_scale=rs))
            print(f"resid ln_mod fp8 {M}x{C}: {ms * 1e3:.1f} us  {M * C * 11 / ms / 1e9:.2f} TB/s")
            del x, y, h, h8
        from omnivideo.modules.engine import build_rope_table
        for M, C, grid in ((20480, 1536, (13, 30, 52)), (75600, 5120, (21, 45, 80))):
            # q and k of one layer live side by side in the fused QKV output (row stride 3C)
            qkv = torch.randn(M, 3 * C, device="cuda").to(BF16)
            wq = torch.randn(C, device="cuda").to(BF16)
            rope = build_rope_table(128, 1024, qkv.device)
            o = torch.empty(M, C, device="cuda", dtype=BF16)
            n = grid[0] * grid[1] * grid[2]
            ms = timeit(lambda: H.rmsnorm_rope(qkv[:, :C], wq, o, 128, 1e-6, rope=rope, n_rope=n, grid=grid))
            print(f"rmsnorm_rope {M}x{C}: {ms * 1e3:.1f} us  {M * C * 4 / ms / 1e9:.2f} TB/s")
            ms = timeit(lambda: H.rmsnorm_rope(qkv[:, :C], wq, o, 128, 1e-6))
            print(f"rmsnorm {M}x{C}: {ms * 1e3:.1f} us  {M * C * 4 / ms / 1e9:.2f} TB/s")
            del qkv, o
    if "lnrot" in args.what:        # the same kernels on rotating buffers: nothing is left in the 256 MB last-level cache
        M, C, sets = 20480, 1536, 8
        xs = [torch.randn(M, C, device="cuda") for _ in range(sets)]
        ys = [torch.randn(M, C, device="cuda").to(BF16) for _ in range(sets)]
        hs = [torch.empty(M, C, device="cuda", dtype=BF16) for _ in range(sets)]
        sh, sc, gt = (torch.randn(C, device="cuda") * 0.1 for _ in range(3))
        def rot():
            for x, y, h in zip(xs, ys, hs):
                H.resid_layernorm_modulate(x, y, h, 1e-6, res_gate=gt, shift=sh, scale=sc)
        ms = timeit(rot) / sets
        print(f"resid ln_mod {M}x{C} rotating: {ms * 1e3:.1f} us  {M * C * 12 / ms / 1e9:.2f} TB/s")
        def rot2():
            for x, h in zip(xs, hs):
                H.layernorm_modulate(x, h, 1e-6, shift=sh, scale=sc)
        ms = timeit(rot2) / sets
        print(f"ln_mod {M}x{C} rotating: {ms * 1e3:.1f} us  {M * C * 6 / ms / 1e9:.2f} TB/s")
        def rot3():
            for y, h in zip(ys, hs):
                h.copy_(y)
        ms = timeit(rot3) / sets
        print(f"torch copy bf16 {M}x{C} rotating: {ms * 1e3:.1f} us  {M * C * 4 / ms / 1e9:.2f} TB/s")
        def rot4():
            for i in range(0, sets, 2):
                xs[i].copy_(xs[i + 1])
        ms = timeit(rot4) / (sets // 2)
        print(f"torch copy f32 {M}x{C} rotating: {ms * 1e3:.1f} us  {M * C * 8 / ms / 1e9:.2f} TB/s")
    if "lnhot" in args.what:        # the rotating LayerNorm right behind a self-attention launch, as in the step
        M, C, sets = 20480, 1536, 8
        xs = [torch.randn(M, C, device="cuda") for _ in range(sets)]
        ys = [torch.randn(M, C, device="cuda").to(BF16) for _ in range(sets)]
        hs = [torch.empty(M, C, device="cuda", dtype=BF16) for _ in range(sets)]
        sh, sc, gt = (torch.randn(C, device="cuda") * 0.1 for _ in range(3))
        q, k, v = (torch.randn(1, 20480, 12, 128, device="cuda").to(BF16) for _ in range(3))
        o = torch.empty_like(q)
        for hot in (False, True):
            tot, n = 0.0, 0
            for it in range(40):
                if hot:
                    H.flash_attn(q, k, v, o, k_len=20280)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for x, y, h in zip(xs, ys, hs):
                    H.resid_layernorm_modulate(x, y, h, 1e-6, res_gate=gt, shift=sh, scale=sc)
                e1.record()
                torch.cuda.synchronize()
                if it >= 10:
                    tot += e0.elapsed_time(e1) / sets
                    n += 1
            print(f"resid ln_mod rotating, {'behind attention' if hot else 'alone'}: {tot / n * 1e3:.1f} us")
    if "gemm14" in args.what:       # the A14B step's Linear shapes
        for N, K, epi in ((15360, 5120, 0), (5120, 5120, 0), (13824, 5120, 1), (5120, 13824, 0)):
            gemm(75600, N, K, epi)
    if "ffnseq" in args.what:       # FFN-up then FFN-down on the step's buffers: the pair, as the step runs it
        M, C, F = 40960, 1536, 8960
        hb = torch.randn(M, C, device="cuda").to(BF16)
        ffn = torch.empty(M, F, device="cuda", dtype=BF16)
        res = torch.empty(M, C, device="cuda", dtype=BF16)
        w0 = (torch.randn(F, C, device="cuda") / 40).to(BF16)
        w2 = (torch.randn(C, F, device="cuda") / 90).to(BF16)
        b0, b2 = torch.zeros(F, device="cuda", dtype=BF16), torch.zeros(C, device="cuda", dtype=BF16)
        def pair():
            H.gemm(hb, w0, ffn, b0, H.EPI_BF16_GELU)
            H.gemm(ffn, w2, res, b2, H.EPI_BF16)
        ms = timeit(pair)
        print(f"FFN-up + FFN-down M{M}: {ms:.3f} ms  {4.0 * M * C * F / ms / 1e9:.0f} TFLOP/s")
    if "ffnchunk" in args.what:     # the same pair in row chunks: the intermediate of a chunk is consumed while still cache-resident
        M, C, F = 40560, 1536, 8960
        hb = torch.randn(M, C, device="cuda").to(BF16)
        ffn = torch.empty(M, F, device="cuda", dtype=BF16)
        res = torch.empty(M, C, device="cuda", dtype=BF16)
        w0 = (torch.randn(F, C, device="cuda") / 40).to(BF16)
        w2 = (torch.randn(C, F, device="cuda") / 90).to(BF16)
        b0, b2 = torch.zeros(F, device="cuda", dtype=BF16), torch.zeros(C, device="cuda", dtype=BF16)
        for rows in (M, 20480, 10240, 8192, 5120):
            def run():
                for r0 in range(0, M, rows):
                    H.gemm(hb[r0:r0 + rows], w0, ffn[r0:r0 + rows], b0, H.EPI_BF16_GELU)
                    H.gemm(ffn[r0:r0 + rows], w2, res[r0:r0 + rows], b2, H.EPI_BF16)
            ms = timeit(run)
            print(f"FFN-up + FFN-down M{M} in chunks of {rows} rows: {ms:.3f} ms  {4.0 * M * C * F / ms / 1e9:.0f} TFLOP/s", flush=True)
    if "gemmdown" in args.what:     # one shape for counter passes: FFN-down of the batched step
        gemm(40960, 1536, 8960, 0)
    if "gemmlib" in args.what:      # the vendor library (torch -> hipBLASLt) on the step's GEMM shapes, for orientation
        M = 40960
        for N, K in ((4608, 1536), (1536, 1536), (8960, 1536), (1536, 8960)):
            a = torch.randn(M, K, device="cuda").to(BF16)
            w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(BF16)
            b = torch.randn(N, device="cuda").to(BF16)
            out = torch.empty(M, N, device="cuda", dtype=BF16)
            ms = timeit(lambda: torch.nn.functional.linear(a, w, b))
            ms2 = timeit(lambda: H.gemm(a, w, out, b, 0))
            print(f"M{M} N{N} K{K}: library {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s | "
                  f"gemm256 {ms2:.3f} ms {2.0 * M * N * K / ms2 / 1e9:.0f} TFLOP/s", flush=True)
    if "lnseq" in args.what:        # LayerNorm in the step's own surroundings: FFN-up, FFN-down, then the norm
        M, C, F = 20480, 1536, 8960
        x = torch.randn(M, C, device="cuda")
        hb = torch.randn(M, C, device="cuda").to(BF16)
        ffn = torch.empty(M, F, device="cuda", dtype=BF16)
        res = torch.empty(M, C, device="cuda", dtype=BF16)
        w0 = (torch.randn(F, C, device="cuda") / 40).to(BF16)
        w2 = (torch.randn(C, F, device="cuda") / 90).to(BF16)
        b0, b2 = torch.zeros(F, device="cuda", dtype=BF16), torch.zeros(C, device="cuda", dtype=BF16)
        sh, sc, gt = (torch.randn(C, device="cuda") * 0.1 for _ in range(3))
        for mode in ("ln only", "gemm gemm ln"):
            tot, n = 0.0, 0
            for it in range(60):
                if mode != "ln only":
                    H.gemm(hb, w0, ffn, b0, H.EPI_BF16_GELU)
                    H.gemm(ffn, w2, res, b2, H.EPI_BF16)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                H.resid_layernorm_modulate(x, res, hb, 1e-6, res_gate=gt, shift=sh, scale=sc)
                e1.record()
                if it % 10 == 9:
                    torch.cuda.synchronize()
                if it >= 20:
                    torch.cuda.synchronize()
                    tot += e0.elapsed_time(e1)
                    n += 1
            print(f"resid ln_mod, {mode}: {tot / n * 1e3:.1f} us (event to event)")
    if "gemm" in args.what:
        M = 40560
        gemm(M, 4608, 1536, 0)      # fused QKV
        gemm(M, 1536, 1536, 3)      # o-proj + gated residual
        gemm(M, 8960, 1536, 1)      # ffn up + GELU
        gemm(M, 1536, 8960, 3)      # ffn down + gated residual
