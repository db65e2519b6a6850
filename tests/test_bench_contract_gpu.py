"""bench.py prints exactly ONE JSON line with the fields the driver's contract names."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def test_bench_contract_tiny_workload():
    env = dict(os.environ)
    env["MPNN_CPU_THREADS"] = "4"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup",
                        "1", "--cpu-seconds", "1", "--side-workloads", "on", "--side-scale", "0.02"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"]
    assert d["forward"]["value"] > d["value"]                      # inference pass is faster than the training step
    assert d["cold_batch"]["total_ms"] > 0                          # CSR + plan build per new batch, beside the step time
    assert d["streaming"]["device_features"]["ms_per_step"] > 0 and d["streaming"]["uploaded_features"]["ms_per_step"] > 0
    if os.environ.get("MPNN_GRU_MATH") != "fp32":                   # (a run that is on the fp32 pipe already has no side leg)
        assert d["fp32_pipe"]["ms_per_step"] > 0                    # strict fp32 MFMA step (child process)
    assert cb["host_cpu_count"] >= cb["cores"] and cb["extrapolated_seconds_for_2k_molecules"] > 0
    assert all(v is None or v > 0 for v in d["kernels_ms"].values())
    # every other BASELINE config stepped in the same run (here at 2 % of its molecules): c1 with the reference's batches of
    # 16 beside the whole-set batch, c3 (attention model), c4, c5
    wl = d["workloads"]
    assert set(wl) == {"c1", "c3", "c4", "c5"}
    for name, w in wl.items():
        assert "error" not in w, (name, w)
        assert w["train_ms"] > w["fwd_ms"] > 0 and w["edges_per_s"] > 0 and w["edges"] > 0
        assert w["dominant_kernel"]["ms_per_step"] > 0 and w["dominant_kernel"]["kernel"] in w["kernels_ms_per_launch"]
    assert (wl["c3"]["hidden"], wl["c4"]["hidden"], wl["c5"]["hidden"]) == (128, 128, 256) and wl["c3"]["mp_steps"] == 5
    assert wl["c1"]["batches_of_16"]["train_edges_per_s"] > 0 and wl["c1"]["batches_of_16"]["batches"] >= 2


def test_batch_stream_hands_over_fresh_batches_one_ahead():
    """mpnn_amd/streaming.py: batches prepared on a side stream by a worker thread give the same step results as batches
    prepared in line, in order, and every batch exactly once."""
    import numpy as np
    import torch
    from mpnn_amd import parallel, synth
    from mpnn_amd.graph import MolGraph
    from mpnn_amd.models.basic_model import BasicModel
    from mpnn_amd.streaming import BatchStream
    dev = torch.device("cuda:0")
    H = 64
    hosts = [synth.make_molecules(300 + 50 * i, H, seed=40 + i) for i in range(5)]
    torch.manual_seed(3)
    model = BasicModel(H, 4, H, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=2).to(dev)
    bucket = parallel.GradientBucket([p for n, p in model.named_parameters() if not n.startswith("of.")])

    def step(feats, g, mask):
        bucket.zero()
        state, _ = model.message_passing(feats, g, g, mask)
        state.backward(gradient=torch.full_like(state, 1e-3))
        return state.detach().clone(), bucket.flat.clone()

    want = []
    for hb in hosts:
        g = MolGraph.from_molbatch(hb, dev).prepare()
        want.append(step(torch.from_numpy(hb.atom_feat).to(dev), g, torch.ones(g.num_nodes, 1, device=dev)))
    bs = BatchStream(hosts, dev, H)
    got = []
    for b in bs:
        got.append(step(b.feats, b.graph, b.mask))
        bs.done_with(b)
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for (s1, g1), (s2, g2) in zip(got, want):
        assert torch.equal(s1, s2)
        assert float((g1 - g2).abs().max()) <= 2e-6 * float(g2.abs().max())
