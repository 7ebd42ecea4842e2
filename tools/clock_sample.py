"""Development aid: runs one workload in a loop for a few seconds while a thread polls rocm-smi for the shader
clock; prints the median / min / max and the time per iteration.  Workloads: vae (cfg2 decode, default), attn
(self-attention at the cfg2 shape), gemm (the FFN-up GEMM of the 1.3B model with its GELU epilogue), gemmqkv.
usage: python tools/clock_sample.py [seconds] [workload]"""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "omni-video_amd"))
import torch  # noqa: E402


def poll(stop, rows):
    while not stop.is_set():
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showclocks", "--showpower"], capture_output=True,
                                 text=True, timeout=10).stdout
        except Exception as e:      # noqa: BLE001
            rows.append(("error", str(e)))
            return
        sclk = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", out)
        pw = re.search(r"Power \(W\):\s*([\d.]+)", out)
        rows.append((int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None, out if not sclk else ""))
        time.sleep(0.3)


def workload(name):
    if name == "vae":
        from omnivideo.modules.vae2_1 import Wan2_1_VAE
        with torch.device("cuda"):
            v = Wan2_1_VAE(z_dim=16, vae_pth=None, device="cuda")
        z = torch.randn(16, 13, 60, 104, device="cuda")
        return lambda: v.decode([z])
    import omnivideo._hip as H
    bf = torch.bfloat16
    if name == "attn":
        B, L, N, D = 2, 20280, 12, 128
        q, k, v = (torch.randn(B, L, N, D, device="cuda").to(bf) for _ in range(3))
        o = torch.empty_like(q)
        return lambda: [H.flash_attn(q, k, v, o) for _ in range(20)]
    M, N, K, epi = (40560, 8960, 1536, 1) if name == "gemm" else (40560, 4608, 1536, 0)
    a = torch.randn(M, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
    b = torch.randn(N, device="cuda").to(bf)
    out = torch.empty(M, N, device="cuda", dtype=bf)
    return lambda: [H.gemm(a, w, out, b, epi) for _ in range(100)]


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    name = sys.argv[2] if len(sys.argv) > 2 else "vae"
    step = workload(name)
    step()
    torch.cuda.synchronize()
    stop, rows = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, rows))
    th.start()
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        step()
        torch.cuda.synchronize()
        n += 1
    dt = (time.time() - t0) / n
    stop.set()
    th.join()
    clk = sorted(r[0] for r in rows if r[0])
    pw = sorted(r[1] for r in rows if len(r) > 1 and r[1])
    print(f"{name}: {dt:.4f} s per iteration x {n}; samples {len(rows)}")
    if clk:
        print(f"sclk MHz median {clk[len(clk) // 2]} min {clk[0]} max {clk[-1]}")
    if pw:
        print(f"power W median {pw[len(pw) // 2]} min {pw[0]} max {pw[-1]}")
    if not clk and rows:
        print("rocm-smi output not parsed:", rows[0][-1][:600])


if __name__ == "__main__":
    main()
