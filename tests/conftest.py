import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "omni-video_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def check_rel(label, value, limit):
    """Model / loop-level parity: prints the observed relative L2 error (pytest -s shows it, and it is
    appended to gpurun_out/parity_observed.jsonl when that directory exists) and asserts it below
    `limit`.  Limits are set to about three times the value observed on MI355X when the test was last
    calibrated (recorded beside each call), so a regression that triples an error fails."""
    import json
    value = float(value)
    print(f"PARITY {label}: rel L2 = {value:.3e} (limit {limit:.1e})")
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_observed.jsonl"), "a") as f:
            f.write(json.dumps({"label": label, "rel_l2": value, "limit": limit}) + "\n")
    assert value < limit, f"{label}: rel L2 {value:.3e} >= {limit:.1e}"
