"""The C++ multi-GPU host example (examples/multi_gpu_batch.cpp): one host thread + one handle per slice of
the batch, results gathered from the handles' device views -- ncclAllGather (RCCL) when every slice has its own device, device-to-device
copies when slices share one (this 1-GPU box with world = 2).  The program itself compares the gathered result bit for bit with ONE
handle solving the whole batch."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "multi_gpu_batch")


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(EXE)


@pytest.mark.parametrize("world,batch,N", [(1, 96, 40), (2, 96, 40), (3, 100, 24), (2, 2048, 100),
                                           (2, 8192, 100)])   # the last: cfg 4 at full size (8192 unicycle OCPs), two slices of 4096 through the instance queue
def test_slices_match_single_handle(world, batch, N):
    p = subprocess.run([EXE, str(world), str(batch), str(N)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout, p.stderr)
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["identical_to_single_handle"] is True and r["world"] == world and r["global_batch"] == batch
    if world == 1:
        assert "ncclAllGather" in r["gather"]     # the RCCL call site runs (a communicator of one rank on this box)
    else:
        assert r["devices"] >= world or "copies" in r["gather"]
