"""Multi-process (gloo, CPU) coverage of the N>1 path: sequence-parallel groups,
all-to-all / gather collectives and the Ulysses sharding arithmetic."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def launch(nproc, sp, port):
    import socket
    with socket.socket() as sk:          # any free port (the parameter only labels the case)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SP_SIZE=str(sp), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "_sp_worker.py")]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("nproc,sp,port", [(2, 2, 29621), (4, 2, 29622), (3, 3, 29623)])
def test_sequence_parallel_collectives(nproc, sp, port):
    res = launch(nproc, sp, port)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count(" ok") == nproc
