"""The DPM-Solver++ oracle (oracle/dpm.py) against the reference scheduler's schedules and
trajectories (tests/golden/dpm.npz, made by oracle/gen_golden.py from
omnivideo/utils/fm_solvers.py)."""
import os

import numpy as np
import pytest
import torch

from omnivideo.utils import synthetic as syn
from oracle import dpm as D


@pytest.fixture(scope="module")
def G(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "dpm.npz")))


def linear_model(lat, t, which):
    a = 0.9 if which == "c" else 0.7
    return (a * lat.float() + 0.001 * float(t) * torch.sin(lat.float() * 3)).float()


@pytest.mark.parametrize("steps,shift", [(20, 5.0), (40, 5.0), (50, 5.0), (50, 12.0)])
def test_schedule(G, steps, shift):
    sig, ts = D.schedule(steps, shift)
    assert np.array_equal(ts.numpy(), G[f"timesteps_{steps}_{shift:g}"])
    assert np.array_equal(sig.numpy(), G[f"sigmas_{steps}_{shift:g}"])
    assert ts[0] == 1000 and sig[0] == 1.0 and sig[-1] == 0.0


@pytest.mark.parametrize("dt_name,dt", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("steps", [20, 3, 1])
def test_trajectory(G, dt_name, dt, steps):
    sched = D.DPMpp(steps, 5.0)
    lat = syn.randn("dpm.noise", (16, 2, 4, 6), 7).to(dt)
    for i, t in enumerate(sched.timesteps):
        c, u = linear_model(lat, t, "c"), linear_model(lat, t, "u")
        lat = sched.step(u + 3.0 * (c - u), lat)
        assert lat.dtype == torch.float32
        assert torch.equal(lat, torch.from_numpy(G[f"traj_{dt_name}_{steps}"][i])), i
