"""HIP-graph recording of a prepared training step (mpnn_amd/capture.py): the cases live in tests/capture_cases_gpu.py and
run in one child process -- an illegal call inside a capture aborts the process instead of raising (rounds 2 and 3 lost a
bench run that way: gpurun_out/gcap.log, b_c1_g.err), and the suite must survive that."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_recorded_training_steps_in_a_child_process():
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "tests/capture_cases_gpu.py"]
    r = subprocess.run(cmd, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-30:])
    assert r.returncode == 0, "recording cases failed (rc %d):\n%s" % (r.returncode, tail)
    last = r.stdout.strip().splitlines()[-1]
    assert "5 passed" in last, tail
