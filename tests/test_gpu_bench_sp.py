"""bench.py's sequence-parallel gate, run the way the driver runs it (this script as N ranks), with the ranks
sharing cuda:0: every sequence-parallel number has to carry `sp_verified`, obtained by comparing a sharded
forward with an unsharded one bit for bit; a damaged exchange has to flip it and withhold the value; a single
peer-mapped communicator that fails falls back to the ordinary one; a configuration that does not fit the
device is refused with the planned figure before anything is allocated.  Shapes: the 17-frame 256x256 workload
of the 1.3B model with 2 layers (the gate's logic does not depend on the depth)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = ["--workload", "cfg1", "--layers", "2", "--parallel", "sp", "--rehearse-on-one-gpu", "--no-vae",
        "--no-cpu-baseline", "--steps", "2", "--warmup", "1", "--companion", "off"]


def run_bench(n, extra, expect_rc=0, timeout=280, **more_env):
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", OVD_PEER_TIMEOUT_S="20", **more_env)
    env.pop("OVD_SP_COMM", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + BASE + extra, env=env,
                         capture_output=True, text=True, timeout=timeout)
    assert (res.returncode == 0) == (expect_rc == 0), res.stdout[-1500:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert lines, res.stdout[-1500:] + res.stderr[-4000:]
    assert "[bench] phase " in res.stderr                  # the trail a killed job would leave behind
    return json.loads(lines[-1]), res


def test_sp_gate_verifies_both_communicators_two_ranks():
    out, _ = run_bench(2, ["--sp-comm", "host,peer"])
    assert out["sp_verified"] is True and out["config"]["parallelism"] == "sp2"
    runs = {r["sp_comm"]: r for r in out["sp_comm_runs"]}
    assert set(runs) == {"host", "peer"} and all(r["sp_verified"] for r in runs.values()), runs
    assert "bitwise equal" in out["sp_verification"]


def test_sp_gate_four_ranks_fine_grained_payload():
    out, _ = run_bench(4, ["--sp-comm", "peer"])
    assert out["sp_verified"] is True and [r["sp_comm"] for r in out["sp_comm_runs"]] == ["peer"]
    assert out["sp_comm_runs"][0]["payload"] == "default"
    out, _ = run_bench(4, ["--sp-comm", "peer"], OVD_PEER_PAYLOAD="fine")
    assert out["sp_verified"] is True and out["sp_comm_runs"][0]["payload"] == "fine"


def test_damaged_exchange_is_caught_and_peer_falls_back():
    out, _ = run_bench(2, ["--sp-comm", "peer", "--test-corrupt-exchange", "peer"])
    runs = out["sp_comm_runs"]
    assert [r["sp_comm"] for r in runs] == ["peer", "host"]
    assert runs[0]["sp_verified"] is False and "DIFFERENT" in runs[0]["verification"]
    assert runs[1]["sp_verified"] is True and out["sp_verified"] is True
    assert "failed verification" in out["sp_comm_fallback"] and out["config"]["sp_comm"] == "host"
    out, _ = run_bench(2, ["--sp-comm", "host", "--test-corrupt-exchange", "all"])
    assert out["sp_verified"] is False and out["value"] is None and "did not verify" in out["error"]


def test_hybrid_data_parallel_of_sequence_parallel_groups():
    """4 ranks = 2 videos x 2-rank sequence-parallel groups (reference omnivideo/distributed/util.py:43-66)."""
    out, _ = run_bench(4, ["--sp-size", "2", "--sp-comm", "host"])
    assert out["sp_verified"] is True and out["config"]["parallelism"] == "dp2 x sp2" and out["scaling"] == "weak"


def test_configuration_that_does_not_fit_is_refused_with_the_planned_figure():
    args = ["--gpus", "4", "--workload", "cfg4", "--parallel", "sp", "--rehearse-on-one-gpu", "--no-vae",
            "--no-cpu-baseline", "--steps", "1", "--warmup", "1", "--companion", "off"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                         timeout=200)
    assert res.returncode != 0
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["value"] is None and "refused" in line["error"] and "GiB per rank" in line["error"], line
