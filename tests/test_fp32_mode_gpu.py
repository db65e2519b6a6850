"""The strict-fp32 math mode (MPNN_GRU_MATH=fp32: every contraction on v_mfma_f32_32x32x2_f32, no operand splits) under the
driver's own `pytest -m gpu` run.

The switch is read once per process (csrc/capi.hip), so the default-mode suite cannot flip it: this test starts ONE
child test runner with the variable set and a representative subset of the parity cases -- GRU forward / backward at
widths 64, 128, 256 and a generic one, message + neighbour sum, the reference fixtures of the basic, lipo and attention
models -- and requires it to pass.  (The fused split-arithmetic kernels skip themselves in that mode; the subset below
is what exercises the fp32 twins.)"""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SUBSET = [
    "tests/test_kernels_gpu.py::test_gru_against_reference_fixture",
    "tests/test_kernels_gpu.py::test_gru_random",
    "tests/test_kernels_gpu.py::test_edge_message",
    "tests/test_backward_gpu.py::test_gru_update_backward",
    "tests/test_backward_gpu.py::test_gru_backward_random",
    "tests/test_backward_gpu.py::test_gru_forward_backward_at_tile_boundaries",
    "tests/test_backward_gpu.py::test_message_aggregate_node",
    "tests/test_backward_gpu.py::test_basic_model_backward",
    "tests/test_backward_gpu.py::test_lipo_model_backward",
    "tests/test_operators_gpu.py::test_basic_model_forward",
    "tests/test_operators_gpu.py::test_lipo_model_forward",
    "tests/test_operators_gpu.py::test_att_edge_network",
    "tests/test_configs_gpu.py::test_basic_model_at_config_width",
    "tests/test_parity_round3_gpu.py::test_att_model_against_reference_fixture",
]


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get("MPNN_GRU_MATH") == "fp32", reason="already the fp32-mode run")
def test_strict_fp32_mode_subset_in_a_child_process():
    env = dict(os.environ, MPNN_GRU_MATH="fp32",
               MPNN_PARITY_OUT=os.path.join(REPO, "gpurun_out", "r04_parity_fp32.json"))
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + SUBSET
    r = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = "\n".join(r.stdout.splitlines()[-25:])
    assert r.returncode == 0, "fp32-mode subset failed:\n" + tail
    last = r.stdout.strip().splitlines()[-1]
    assert " passed" in last and "failed" not in last, tail
    passed = int(last.split(" passed")[0].split()[-1])
    assert passed >= 60, tail                               # (a subset that silently shrank to nothing is not evidence)
    print("fp32-mode child run: " + last)
