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
