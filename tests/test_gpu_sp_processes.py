"""The sequence-parallel surface run by real processes: 2 fresh child ranks on cuda:0 (gloo + the
host-staged communicator, because two RCCL ranks cannot share one GPU) execute
`OmniVideoX2XUnified(use_usp=True, sp_size=2).generate(...)` - group creation, `_apply_usp`'s method
binding, the context broadcast, the sharded DiT forward, the row-sharded VAE decode - and the
reference-named `sp_attn_forward`, `sp_cross_attn_forward`, `distributed_attention` on row shards.
Every result must equal the single-process run bit for bit (reference: x2x_gen_unified.py:172-182,
375-380; full_sequence_parallel.py:63-232; ulysses.py:8-56).  A third child runs the RCCL
communicator itself on a 1-rank "nccl" group."""
import os
import subprocess
import sys

import pytest
import torch

import _sp_case as case

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch(script, args, nproc, timeout=300, **extra_env):
    port = free_port()
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, script)] + list(args)
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-6000:]
    assert res.stdout.count(" ok") == nproc, res.stdout[-2000:]


@pytest.mark.parametrize("sp_comm", ["auto", "peer", "peer-pull"])
def test_generate_with_usp_two_processes_equals_single_process(tmp_path, sp_comm):
    """sp_comm "auto": the host-staged communicator (collectives eager between graphs); "peer": the
    peer-mapped communicator - the two processes map each other's exchange buffers, the RMSNorm+RoPE kernel
    writes q, k, v into the peer's attention operands, the head-gather reads the peer's attention output,
    and every step, exchanges included, replays as one graph; "peer-pull": the same communicator with
    every exchange as barrier + all-to-all copy kernel."""
    env = dict(OVD_SP_COMM="peer", OVD_PEER_DIRECT="0" if sp_comm == "peer-pull" else "1") if sp_comm != "auto" \
        else dict(OVD_SP_COMM="auto")
    launch("_sp_gpu_worker.py", ["generate", str(tmp_path)], 2, **env)
    with torch.no_grad():
        pipe, cfg = case.build_pipeline(use_usp=False)
        lat, video = case.run_generate(pipe, cfg)
        kw = case.generate_inputs(cfg)
        dit = pipe.high_noise_model.wan_model(kw["noise"].cuda()[None].unbind(0), torch.tensor([640.0], device="cuda"),
                                              [kw["precomputed_context"].cuda()], 72)[0]
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert r1["lat"] is None and r1["video"] is None            # only the group's first rank returns
    assert torch.equal(r0["lat"], lat.cpu())
    assert tuple(r0["video"].shape) == (3, 9, 64, 96) and torch.equal(r0["video"], video.cpu())
    for r in (r0, r1):                                            # sp_dit_forward: full output on every rank
        assert torch.equal(r["dit"], dit.cpu())


def test_sp_attention_forwards_two_processes_equal_unsharded(tmp_path):
    launch("_sp_gpu_worker.py", ["attention", str(tmp_path)], 2)
    with torch.no_grad():
        pipe, cfg = case.build_pipeline(use_usp=False)
        blk = pipe.high_noise_model.wan_model.blocks[1]
        x, ctx, grid, seq_lens = case.attention_inputs(cfg)
        want_self = blk.self_attn(x, seq_lens, grid).cpu()
        want_cross = blk.cross_attn(x, ctx, None).cpu()
    s = x.shape[1] // 2
    for r in range(2):
        got = torch.load(tmp_path / f"rank{r}.pt", weights_only=True)
        assert torch.equal(got["self"], want_self[:, r * s:(r + 1) * s]), f"sp_attn_forward rank {r}"
        assert torch.equal(got["cross"], want_cross[:, r * s:(r + 1) * s]), f"sp_cross_attn_forward rank {r}"
        assert torch.equal(got["ulysses"], got["ulysses_ref_rows"]), f"distributed_attention rank {r}"


def test_peer_mapped_collectives_two_processes(tmp_path):
    launch("_sp_gpu_worker.py", ["peer", str(tmp_path)], 2, timeout=120)


def test_peer_mapped_engine_forward_four_processes_equals_unsharded(tmp_path):
    """4 ranks x 1 head each, 80 rows with 8 padding rows: every rank's full output equals the unsharded
    forward bit for bit, with the kernels addressing the peers directly (one guidance branch's exchange on a
    second stream beside the other's attention, and the serial order) and with the copy-kernel exchange,
    bf16 and reduced precision, eager and replayed as one graph."""
    import _sp_gpu_worker as W
    launch("_sp_gpu_worker.py", ["peer_engine", str(tmp_path)], 4, timeout=240)
    with torch.no_grad():
        want = {p: W.engine_case(p).cpu() for p in ("bf16", "fp8")}
    for r in range(4):
        got = torch.load(tmp_path / f"rank{r}.pt", weights_only=True)
        assert sorted(got) == ["bf16_direct", "bf16_direct_serial", "bf16_pull", "fp8_direct", "fp8_direct_serial", "fp8_pull"]
        for name, t in got.items():
            assert torch.equal(t, want[name.split("_")[0]]), f"rank {r} {name}"


def test_peer_mapped_a14b_geometry_eight_processes_equals_unsharded(tmp_path):
    """The sequence-parallel configuration of the 81-frame 720p workload at its real per-rank shapes: one
    layer of the A14B width over 8 ranks (8 processes sharing this GPU), q | k | v written into the eight
    peers' attention operands, outputs read back from them, replayed as a graph: bitwise equal to the
    unsharded forward, in bf16 and in the reduced-precision mode."""
    import _sp_gpu_worker as W
    launch("_sp_gpu_worker.py", ["peer_a14b", str(tmp_path)], 8, timeout=280)
    for precision in ("bf16", "fp8"):
        with torch.no_grad():
            want = W.engine_case(precision, shape="a14b").cpu()
        for r in (0, 7):
            got = torch.load(tmp_path / f"rank{r}.pt", weights_only=True)[precision]
            assert torch.equal(got, want), f"{precision} rank {r}: max diff {(got - want).abs().max().item()}"


def test_rccl_communicator_single_rank_group(tmp_path):
    launch("_nccl_worker.py", [str(tmp_path)], 1)
