"""Data parallelism on the HIP path itself: 2 ranks (gloo, both on cuda:0) run BasicModel on their molecule shards,
sum gradients through parallel.GradientBucket, and must reproduce the single-process gradient of the whole batch."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu

H, T, G = 64, 3, 600


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dev):
    from mpnn_amd.models.basic_model import BasicModel
    torch.manual_seed(23)
    model = BasicModel(H, 4, H, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={},
                       message_steps=T).to(dev)
    with torch.no_grad():
        for mod in model.mf.edge_map.modules():
            if isinstance(mod, torch.nn.Linear):
                torch.nn.init.kaiming_uniform_(mod.weight, nonlinearity="relu")
        last = model.mf.edge_map[-1]
        last.weight.mul_(1e-3)
    return model


def _loss_grad(model, mb, dev, total_mols):
    """Gradient of (1/total_mols) * sum over this batch's molecules of sum(node_state^2) into a GradientBucket."""
    from mpnn_amd import parallel
    from mpnn_amd.graph import MolGraph
    hot = [p for n, p in model.named_parameters() if not n.startswith("of.")]
    bucket = parallel.GradientBucket(hot)
    bucket.zero()
    g = MolGraph.from_molbatch(mb, dev)
    afm = torch.from_numpy(mb.atom_feat).to(dev)
    state, _ = model.message_passing(afm, g, g, torch.ones(mb.num_atoms, 1, device=dev))
    ((state * state).sum() / total_mols).backward()
    return bucket


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from mpnn_amd import parallel, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    mb = synth.make_molecules(G, H, seed=41)
    per_mol = np.add.reduceat(np.diff(mb.row_ptr), mb.atom_ptr[:-1])
    ids = parallel.shard_by_edges(per_mol, world)[rank]
    total = parallel.global_count(len(ids), dev)
    bucket = _loss_grad(_model(dev), synth.select(mb, ids), dev, total)
    flat = bucket.all_reduce().detach().cpu().numpy()
    if rank == 0:
        out.put((flat, total))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hip_gradient_equals_single_process():
    from mpnn_amd import synth
    dev = torch.device("cuda:0")
    mb = synth.make_molecules(G, H, seed=41)
    ref = _loss_grad(_model(dev), mb, dev, float(G)).flat.detach().cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, total = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert total == G
    scale = max(1e-6, float(np.abs(ref).max()))
    assert np.abs(got - ref).max() / scale < 2e-5        # same terms, summed in two parts


def _att_model(dev, sync):
    from mpnn_amd.models.att_model import BasicModel as AttModel
    from mpnn_amd.mpnn_functions import GraphLevelOutput
    torch.manual_seed(29)
    model = AttModel(128, 4, 128, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=3,
                     readout_func=GraphLevelOutput).to(dev)
    model.bn.sync_stats = sync
    return model


def _att_loss_grad(model, mb, dev, total_mols, rows):
    """rows = the GLOBAL atom ids of this batch's atoms: the cotangent is a fixed function of (global atom, column), so
    shards and the whole batch differentiate the same loss (a sum of squares would be constant under the norm)."""
    from mpnn_amd import parallel
    from mpnn_amd.graph import MolGraph
    hot = [p for n, p in model.named_parameters() if not n.startswith("of.")]
    bucket = parallel.GradientBucket(hot)
    bucket.zero()
    g = MolGraph.from_molbatch(mb, dev)
    afm = torch.from_numpy(mb.atom_feat).to(dev)
    state, _ = model.message_passing(afm, g, g, torch.ones(mb.num_atoms, 1, device=dev))
    r = torch.from_numpy(np.asarray(rows, dtype=np.float64)).to(dev)
    cot = torch.sin(0.37 * r[:, None] + 1.3 * torch.arange(128, device=dev, dtype=torch.float64)[None, :]).float()
    ((state * cot).sum() / total_mols).backward()
    return bucket, state.detach()


def _att_worker(rank, world, port, out):
    import torch.distributed as dist
    from mpnn_amd import parallel, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    mb = synth.make_molecules(400, 128, seed=43)
    per_mol = np.add.reduceat(np.diff(mb.row_ptr), mb.atom_ptr[:-1])
    ids = np.sort(parallel.shard_by_edges(per_mol, world)[rank])
    model = _att_model(dev, True)
    assert model._norm_fusable(torch.zeros(1, 128, device=dev))
    rows = np.concatenate([np.arange(mb.atom_ptr[i], mb.atom_ptr[i + 1]) for i in ids])
    bucket, state = _att_loss_grad(model, synth.select(mb, ids), dev, 400.0, rows)
    flat = bucket.all_reduce().detach().cpu().numpy()
    out.put((rank, flat, rows, state.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_fused_norm_chain_with_synced_statistics():
    """The attention model at hidden 128 sharded by graph over 2 ranks with `bn.sync_stats`: the fused update + norm chain
    all-reduces its moments and backward sums, so node states and the summed gradient equal the single-process run over
    the whole batch (the norms couple every atom of the batch)."""
    from mpnn_amd import ops, synth
    dev = torch.device("cuda:0")
    if not ops.gru_norm_applies(128, torch.zeros(1, device=dev)):
        pytest.skip("the fused update + norm kernels are split-precision kernels (not under MPNN_GRU_MATH=fp32)")
    mb = synth.make_molecules(400, 128, seed=43)
    ref_bucket, ref_state = _att_loss_grad(_att_model(dev, False), mb, dev, 400.0, np.arange(mb.num_atoms))
    ref = ref_bucket.flat.detach().cpu().numpy()
    ref_state = ref_state.cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_att_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    scale = max(1e-6, float(np.abs(ref).max()))
    for rank, flat, rows, state in got:
        assert np.abs(state - ref_state[rows]).max() < 5e-5, rank
        assert np.abs(flat - ref).max() / scale < 1e-4, rank


def _run_bench(extra):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1",
                        "--no-cpu"] + extra, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    """`bench.py --gpus 2` without a launcher must report n_gpus 2 (it starts its ranks as a child torchrun; on this
    one-GPU box they share the device over gloo and the line says so)."""
    d = _run_bench(["--gpus", "2"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp2"
    sh = d["sharding"]
    assert sh["global_mols"] == 4000 and sh["allreduce_ms"] is not None
    if torch.cuda.device_count() < 2:
        assert "rehearsal" in d


def test_bench_strong_scaling_partitions_one_global_set():
    one = _run_bench(["--scaling", "strong"])
    two = _run_bench(["--scaling", "strong", "--gpus", "2"])
    assert one["scaling"] == two["scaling"] == "strong"
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["sharding"]["global_mols"] == two["sharding"]["global_mols"] == 16000
    assert one["sharding"]["global_edges"] == two["sharding"]["global_edges"]            # the same molecules
    assert two["sharding"]["edges_per_rank_max"] - two["sharding"]["edges_per_rank_min"] <= 200
    assert one["config"]["micro_batches_per_step"] == 8 and two["config"]["micro_batches_per_step"] == 4
