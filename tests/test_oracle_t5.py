"""The umT5 encoder oracle (oracle/t5.py) against outputs of the reference's own T5Encoder
(tests/golden/t5_tiny.npz, made by oracle/gen_golden.py from omnivideo/modules/t5.py)."""
import os

import numpy as np
import pytest
import torch

from omnivideo.utils import synthetic as syn
from oracle import t5 as T

CASES = {"short": (2, 40, [40, 23]), "long": (1, 300, [277])}


@pytest.fixture(scope="module")
def G(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "t5_tiny.npz")))


@pytest.mark.parametrize("name", list(CASES))
def test_fp32_flow_is_bit_exact(G, name):
    cfg = syn.t5_config("tiny")
    sd = syn.make_t5_state_dict(cfg, 0)
    ids, mask = syn.make_t5_ids(cfg, *CASES[name], seed=3)
    assert torch.equal(T.encode(ids, mask, sd, cfg, "fp32"), torch.from_numpy(G[f"{name}_fp32"]))


@pytest.mark.parametrize("name", list(CASES))
def test_bf16_flow(G, name):
    """Every elementwise rounding point is reproduced exactly (checked op by op while writing the
    oracle: norm, linear, cube/GELU, scores, softmax, residual adds).  What remains against the
    reference run on CPU is the summation order inside its bf16 GEMMs: sums that land on a bf16
    rounding tie move by one ulp and propagate - a few percent of the outputs by <= a few ulps."""
    cfg = syn.t5_config("tiny")
    sd = T.round_weights_bf16(syn.make_t5_state_dict(cfg, 0))
    ids, mask = syn.make_t5_ids(cfg, *CASES[name], seed=3)
    got, ref = T.encode(ids, mask, sd, cfg, "bf16"), torch.from_numpy(G[f"{name}_bf16"])
    assert float((got - ref).norm() / ref.norm()) < 6e-3
    assert float((got != ref).float().mean()) < 0.15


def test_relative_position_buckets():
    b = T.relative_position_bucket(200, 200, 32)
    assert b.min() == 0 and b.max() == 31
    assert b[5, 5] == 0 and b[5, 9] == 16 + 4 and b[9, 5] == 4       # exact range, sign bit = 16
    assert b[0, 199] == 31 and b[199, 0] == 15                        # clipped at max distance
    assert torch.equal(b[3:, 3:], b[:-3, :-3])                        # depends on j - i only


def test_mask_drops_padded_keys():
    """A padded key gets the dtype's lowest value as its score (t5.py:104-109): the valid rows of a
    padded batch entry equal the rows of the same sequence encoded without padding."""
    cfg = syn.t5_config("tiny")
    sd = syn.make_t5_state_dict(cfg, 0)
    ids, mask = syn.make_t5_ids(cfg, 1, 30, [17], seed=4)
    full = T.encode(ids, mask, sd, cfg, "fp32")[0, :17]
    alone = T.encode(ids[:, :17], mask[:, :17], sd, cfg, "fp32")[0]
    torch.testing.assert_close(full, alone, rtol=1e-5, atol=1e-5)
