"""The data-parallel layer on CPU: 2 ranks over gloo must reproduce the single-process gradient."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mpnn_amd import parallel, synth


def test_shard_by_edges_is_a_balanced_partition():
    mb = synth.make_molecules(3000, 4, seed=11, dist="skewed")
    edges = np.diff(mb.row_ptr)[..., None].sum(-1)              # per atom
    per_mol = np.add.reduceat(np.diff(mb.row_ptr), mb.atom_ptr[:-1])
    for world in (1, 2, 4, 8):
        shards = parallel.shard_by_edges(per_mol, world)
        allm = np.concatenate(shards)
        assert np.array_equal(np.sort(allm), np.arange(3000))   # exact partition
        loads = np.array([per_mol[s].sum() for s in shards])
        assert loads.max() - loads.min() <= per_mol.max() + world
    big = parallel.shard_by_edges(np.random.default_rng(0).integers(10, 400, 80_000), 8)   # serpentine branch
    loads = np.array([len(s) for s in big])
    assert loads.max() - loads.min() <= 1
    assert np.array_equal(np.sort(np.concatenate(big)), np.arange(80_000))
    del edges


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))


def _worker(rank, world, port, xs, ys, shards, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _toy()
    bucket = parallel.GradientBucket(model.parameters())
    ids = torch.from_numpy(shards[rank])
    total = parallel.global_count(len(ids), torch.device("cpu"))
    bucket.zero()
    loss = ((model(xs[ids]) - ys[ids]) ** 2).sum() / total        # this shard's part of the global mean
    loss.backward()
    flat = bucket.all_reduce().clone()
    if rank == 0:
        out.put(flat.numpy())
    dist.destroy_process_group()


def test_two_rank_gradient_equals_single_process():
    G = 37
    g = torch.Generator().manual_seed(3)
    xs, ys = torch.rand(G, 6, generator=g), torch.rand(G, 1, generator=g)
    shards = parallel.shard_by_edges(np.arange(G) % 7 + 1, 2)
    model = _toy()
    ((model(xs) - ys) ** 2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, xs, ys, shards, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.abs(got - ref).max() < 1e-6


def _bn_worker(rank, world, port, x, mask, w, b, cot, rows, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for name, kw in (("bn", dict(masked_mean=False, eps_inside=True, eps=1e-6)),
                     ("bn1d", dict(masked_mean=True, eps_inside=False, eps=1e-5))):
        ids = rows[rank]
        xs = x[ids].clone().requires_grad_(True)
        ws, bs = (w.clone().requires_grad_(True), b.clone().requires_grad_(True)) if name == "bn1d" else (None, None)
        y, mean, var = parallel.synced_masked_batch_norm(xs, mask[ids], ws, bs, **kw)
        (y * cot[ids]).sum().backward()
        res[name] = (y.detach().numpy(), xs.grad.numpy(), mean.detach().numpy(), var.detach().numpy(),
                     None if ws is None else ws.grad.numpy())
    out.put((rank, res))
    dist.destroy_process_group()


def test_synced_masked_batch_norm_matches_single_process():
    """Statistics all-reduced over 2 gloo ranks == the oracle's masked norms on the whole batch: outputs, moments,
    and gradients wrt the inputs (the weight gradient is each rank's share, summed by the gradient all-reduce)."""
    from oracle import dense_ref as O
    torch.manual_seed(0)
    V, F = 91, 6
    x = torch.randn(V, F)
    mask = (torch.rand(V, 1) > 0.2).float()
    x = x * mask
    w, b = torch.rand(F) + 0.5, torch.rand(F) - 0.5
    cot = torch.randn(V, F)
    rows = [torch.arange(0, 40), torch.arange(40, V)]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, x, mask, w, b, cot, rows, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(out.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    # references on the whole batch
    xr = x.clone().requires_grad_(True)
    ref = O.mask_bn(xr, mask)
    (ref * cot).sum().backward()
    y = np.concatenate([got[0]["bn"][0], got[1]["bn"][0]])
    gx = np.concatenate([got[0]["bn"][1], got[1]["bn"][1]])
    assert np.abs(y - ref.detach().numpy()).max() < 1e-5
    assert np.abs(gx - xr.grad.numpy()).max() < 1e-5
    params = {"weight": w.clone().requires_grad_(True), "bias": b.clone().requires_grad_(True)}
    xr2 = x.clone().requires_grad_(True)
    ref1 = O.mask_bn1d(xr2, mask, params["weight"], params["bias"], training=True)[0]
    (ref1 * cot).sum().backward()
    y1 = np.concatenate([got[0]["bn1d"][0], got[1]["bn1d"][0]])
    gx1 = np.concatenate([got[0]["bn1d"][1], got[1]["bn1d"][1]])
    assert np.abs(y1 - ref1.detach().numpy()).max() < 1e-5
    assert np.abs(gx1 - xr2.grad.numpy()).max() < 2e-5
    assert np.abs(got[0]["bn1d"][4] + got[1]["bn1d"][4] - params["weight"].grad.numpy()).max() < 2e-5
    assert np.abs(got[0]["bn1d"][2] - got[1]["bn1d"][2]).max() == 0          # both ranks hold the same moments


def test_strong_scaling_shard_is_a_partition_of_one_global_set():
    """parallel.strong_scaling_shard: whatever the world size, the ranks' micro-batches hold every molecule of the
    SAME global set exactly once, balanced by edge count, with features keyed on the global molecule id."""
    ref_keys = None
    for world in (1, 2, 4):
        keys, edges, loads = [], 0, []
        for r in range(world):
            micro, info = parallel.strong_scaling_shard(4, 300, r, world, micro_mols=250)
            assert info["global_mols"] == 1200
            assert all(m.num_mols <= 250 for m, _ in micro)
            assert sum(m.num_mols for m, _ in micro) == info["local_mols"]
            for m, k in micro:
                assert k.shape[0] == m.num_atoms and m.atom_feat.shape == (m.num_atoms, 0)
                assert int(m.col_idx.max()) < m.num_atoms and m.row_ptr[-1] == m.num_edges
            keys.append(np.concatenate([k for _, k in micro]))
            loads.append(info["local_edges"])
            edges += info["local_edges"]
        allk = np.sort(np.concatenate(keys))
        assert np.unique(allk).shape[0] == allk.shape[0] and edges == info["global_edges"]
        assert max(loads) - min(loads) <= 120
        if ref_keys is None:
            ref_keys = allk
        assert np.array_equal(allk, ref_keys)
    f = synth.hashed_features(ref_keys[:500], 16)
    assert f.shape == (500, 16) and float(f.abs().max()) <= 1.0 and abs(float(f.mean())) < 0.05
    assert torch.equal(f[:7], synth.hashed_features(ref_keys[:7], 16))
