"""The allocation-free attention entry point (ovd_flash_attn_static_fwd_bf16: tail-balancing workspace
in a device array of the library) against the entry point it is a variant of
(ovd_flash_attn_ws_fwd_bf16 with a caller-owned workspace), through the C-ABI: bit-identical outputs
on two calls in a row, the second on a smaller input (the array still holds the first call's
partial results), then on a shape that does not split at all."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _qkv(B, L, N, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = torch.randn(B, L, 3 * N * 128, device="cuda", generator=g).to(BF16)
    return tuple(qkv[:, :, i * N * 128:(i + 1) * N * 128].unflatten(2, (N, 128)) for i in range(3))


def _both(H, B, L, N, k_len, copies, seed):
    q, k, v = _qkv(B, L, N, seed)
    ws_bytes = H.flash_attn_workspace_bytes(B, L, N, k_len)
    ref = torch.full((B, L, N, 128), float("nan"), device="cuda", dtype=BF16)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device="cuda")
    H.flash_attn(q, k, v, ref, k_len=k_len, last_key_copies=copies, workspace=ws)
    out = torch.full((B, L, N, 128), float("nan"), device="cuda", dtype=BF16)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    H.flash_attn(q, k, v, out, k_len=k_len, last_key_copies=copies, static_workspace=True)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    return ref, out, ws_bytes, free0 - free1


def test_static_workspace_attention_matches_caller_workspace_entry():
    from omnivideo import _hip as H
    H.flash_attn(*_qkv(1, 256, 1, 0), torch.empty(1, 256, 1, 128, device="cuda", dtype=BF16),
                 static_workspace=True)          # first call on the device: address lookup
    torch.cuda.synchronize()
    # 1) the largest split there is on 256 CUs: 2 x 12 heads x 80 query blocks = 7.5 rounds
    ref, out, ws_bytes, grew = _both(H, 2, 20280, 12, 20280, 1, 1)
    assert ws_bytes == 34603008, "expected the shape that fills the built-in array"
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    assert grew == 0, f"device memory in use grew by {grew} bytes during the call"
    # 2) smaller input right after it: fewer split blocks, masked key tail, a weighted last key
    ref, out, ws_bytes, grew = _both(H, 2, 10000, 4, 9001, 37, 2)
    assert 0 < ws_bytes < 34603008
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    assert grew == 0
    # 3) a shape that does not split: the array is not touched, same result as the plain entry
    ref, out, ws_bytes, grew = _both(H, 1, 700, 3, 650, 1, 3)
    assert ws_bytes == 0
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


def test_static_workspace_attention_rejects_bad_arguments():
    from omnivideo import _hip as H
    q, k, v = _qkv(1, 256, 1, 4)
    out = torch.empty(1, 256, 1, 128, device="cuda", dtype=BF16)
    with pytest.raises(RuntimeError, match="k_len"):
        H.flash_attn(q, k, v, out, k_len=0, static_workspace=True)
    with pytest.raises(ValueError, match="either"):
        H.flash_attn(q, k, v, out, workspace=torch.empty(16, dtype=torch.uint8, device="cuda"), static_workspace=True)


def test_static_workspace_rejects_a_split_beyond_the_array_on_a_larger_device():
    """The built-in array covers every split the plan produces on up to 256 compute units.  On a part with more
    (planned here for 304 through OVD_DEVICE_CUS, in a child process: the count is read once) a shape whose last
    round splits 17 query blocks per XCD needs more than the array holds: the allocation-free entry refuses it with
    the figures, the caller-workspace entry takes it."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from omnivideo import _hip as H\n"
        "B, L, N = 1, 14080, 8\n"
        "g = torch.Generator(device='cuda').manual_seed(5)\n"
        "q, k, v = (torch.randn(B, L, N, 128, device='cuda', generator=g).to(torch.bfloat16) for _ in range(3))\n"
        "out = torch.empty_like(q)\n"
        "need = H.flash_attn_workspace_bytes(B, L, N, L)\n"
        "assert need > 34603008, need\n"
        "try:\n"
        "    H.flash_attn(q, k, v, out, static_workspace=True)\n"
        "except RuntimeError as err:\n"
        "    assert 'built-in array holds 34603008' in str(err) and str(need) in str(err), str(err)\n"
        "else:\n"
        "    raise AssertionError('accepted a split the array cannot hold')\n"
        "ws = torch.empty(need, dtype=torch.uint8, device='cuda')\n"
        "H.flash_attn(q, k, v, out, workspace=ws)\n"
        "ref = torch.empty_like(q)\n"
        "H.flash_attn(q, k, v, ref)\n"
        "torch.cuda.synchronize()\n"
        "assert (out.float() - ref.float()).abs().max().item() < 2e-2\n"
        "print('ok')\n"
    ) % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "omni-video_amd")
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OVD_DEVICE_CUS="304"), capture_output=True,
                         text=True, timeout=120)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-500:] + res.stderr[-2000:]
