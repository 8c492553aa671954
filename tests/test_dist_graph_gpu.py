"""The captured training iteration of SEVERAL ranks on the MI355X, exercised with a forced one-rank process group (RCCL
communicators, SyncBN sums and gradient sums in place; MNK_DIST_FORCE=1): tests/dist_graph_worker.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_captured_iteration_with_overlapped_exchange_equals_eager():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MNK_DIST_FORCE="1", MNK_GRAD_OVERLAP="force", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(here, "dist_graph_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST-GRAPH-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
