"""The sampler oracle (oracle/unipc.py) against the reference scheduler's schedules
and trajectories (tests/golden/unipc.npz, made by oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from omnivideo.utils import synthetic as syn
from oracle import unipc as U


@pytest.fixture(scope="module")
def G(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "unipc.npz")))


def linear_model(lat, t, which):
    a = 0.9 if which == "c" else 0.7
    return (a * lat.float() + 0.001 * float(t) * torch.sin(lat.float() * 3)).float()


@pytest.mark.parametrize("steps,shift", [(20, 5.0), (40, 5.0), (50, 5.0), (50, 12.0)])
def test_schedule(G, steps, shift):
    sig, ts = U.schedule(steps, shift)
    assert np.array_equal(ts.numpy(), G[f"timesteps_{steps}_{shift:g}"])
    assert np.array_equal(sig.numpy(), G[f"sigmas_{steps}_{shift:g}"])


def test_expert_boundary_counts():
    # timesteps >= 875 go to the high-noise expert (x2x_gen_unified.py:235,407)
    for steps, n_high in ((50, 21), (40, 17), (20, 9)):
        _, ts = U.schedule(steps, 5.0)
        assert int((ts >= 875).sum()) == n_high


@pytest.mark.parametrize("dt_name,dt", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("steps", [20, 3])
def test_trajectory(G, dt_name, dt, steps, monkeypatch):
    # the goldens were made on CPU, where torch rounds a host scalar to bf16 before
    # multiplying a bf16 tensor; in that mode the oracle must agree bit for bit
    monkeypatch.setattr(U, "SCALAR_MODE", "cpu")
    sched = U.UniPC(steps, 5.0)
    lat = syn.randn("unipc.noise", (16, 2, 4, 6), 7).to(dt)
    for i, t in enumerate(sched.timesteps):
        c, u = linear_model(lat, t, "c"), linear_model(lat, t, "u")
        lat = sched.step(u + 3.0 * (c - u), lat)
        ref = torch.from_numpy(G[f"traj_{dt_name}_{steps}"][i])
        if dt == torch.float32:
            torch.testing.assert_close(lat, ref, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(lat.float(), ref), i
