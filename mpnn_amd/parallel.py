"""Data parallelism over molecules: shard by graph, one flat gradient all-reduce per step.

The reference has no distributed code at all (no torch.distributed / DataParallel call
sites); this layer is specified by BASELINE.json: one process per GPU, molecules partitioned by
graph (no cross-GPU edges, so the message/aggregate/update path needs NO collective), gradients
summed with a single RCCL all-reduce (`torch.distributed` backend "nccl" on ROCm) over xGMI.

Exactness: with loss = (1/G_total) * sum_g loss_g, each rank back-propagates
(1/G_total) * sum_{g in shard} loss_g, and the SUM all-reduce of the gradients equals the
single-process gradient (up to fp32 summation order).  Batch-coupled layers (MaskBatchNorm*)
would need their statistics reduced too; BasicModel has none.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_by_edges(edge_counts, world_size):
    """Longest-processing-time bin packing of molecules by directed-edge count.

    Returns a list of `world_size` int64 arrays of molecule ids (each sorted ascending so a shard
    keeps the batch's molecule order).  Deterministic; every molecule lands on exactly one rank.
    """
    edge_counts = np.asarray(edge_counts, dtype=np.int64)
    G = edge_counts.shape[0]
    if world_size == 1:
        return [np.arange(G, dtype=np.int64)]
    order = np.argsort(-edge_counts, kind="stable")
    if G > 50_000:
        # large batches: dealing the size-sorted list round-robin in serpentine order is within a few
        # edges of LPT and is O(G)
        pos = np.arange(G)
        lap, slot = pos // world_size, pos % world_size
        rank_of = np.where(lap % 2 == 0, slot, world_size - 1 - slot)
        return [np.sort(order[rank_of == r]) for r in range(world_size)]
    load = np.zeros(world_size, dtype=np.int64)
    bins = [[] for _ in range(world_size)]
    for g in order:
        r = int(np.argmin(load))
        bins[r].append(int(g))
        load[r] += edge_counts[g] + 1
    return [np.sort(np.asarray(b, dtype=np.int64)) for b in bins]


class GradientBucket:
    """All trainable parameters' gradients viewed as ONE flat fp32 buffer (a single all-reduce).

    ~4.4 M floats (17.6 MB) at hidden=128: far below the size where xGMI's per-link bandwidth
    matters, so one collective per step is the cheapest schedule (no bucketing/overlap needed).
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:                      # gradients become views into the flat buffer
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        return self.flat


def global_count(local_count, device, group=None):
    """Sum of a per-rank count (e.g. molecules in the shard) over all ranks."""
    t = torch.tensor([float(local_count)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return float(t.item())


def synced_masked_batch_norm(x, mask, weight=None, bias=None, eps=1e-5, masked_mean=True, eps_inside=False, group=None):
    """Masked batch norm whose statistics span ALL ranks of `group` (the lipo / attention models' norms take
    their moments over every atom of the batch, so a batch sharded by graph needs the three sums all-reduced to keep
    single-process results).  Same arithmetic as models/mask_batch_norm.py:5-38 -- mean numerator masked or not,
    variance around the mean, eps inside or outside the root -- written in differentiable torch ops with the
    collective inside the autograd graph (its backward is another all-reduce).  Returns (y, mean, var).
    Without an initialised process group (or world size 1) it is the single-process formula."""
    from torch.distributed.nn.functional import all_reduce as ar
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    mk = mask.reshape(-1, 1)
    y = x.reshape(-1, x.shape[-1])
    s = torch.cat([mk.sum().reshape(1), ((y * mk) if masked_mean else y).sum(dim=0)])
    if multi:
        s = ar(s, group=group)
    cnt, mean = s[0], s[1:] / s[0]
    c = (y - mean) * mk
    q = (c * c).sum(dim=0)
    if multi:
        q = ar(q, group=group)
    var = q / cnt
    scale = (var + eps).sqrt() if eps_inside else var.sqrt() + eps
    if masked_mean:                      # MaskBatchNorm1d: affine on the unmasked normalised value, mask last
        out = (y - mean) / scale
        if weight is not None:
            out = weight * out + bias
        out = out * mk
    else:                                # MaskBatchNorm: mask inside, no parameters
        out = c / scale
    return out.view(x.shape), mean, var


def strong_scaling_shard(num_chunks, chunk_mols, rank, world, seed=317, dist_name="drug", edge_features=4,
                         micro_mols=None):
    """This rank's share of ONE fixed global set of num_chunks * chunk_mols synthetic molecules (BASELINE configs[3]:
    8 x 125k = 1 M), for strong-scaling runs: the same molecules at every world size, partitioned by graph.

    The set is generated chunk by chunk (chunk c = synth.make_molecules(chunk_mols, seed=seed + c), topology only) so
    no rank ever holds 1 M molecules' features; `shard_by_edges` over ALL molecules' edge counts gives every rank a
    balanced list of global molecule ids; the rank's molecules are gathered chunk by chunk and cut into micro-batches
    of at most `micro_mols` molecules (default chunk_mols: what one GPU holds resident for one forward + backward).

    Returns (micro_batches, info): micro_batches = [(MolBatch without features, atom keys int64 for
    synth.hashed_features)], info = dict(global_mols, global_edges, local_mols, local_edges)."""
    from . import synth
    micro_mols = int(micro_mols or chunk_mols)
    chunks = [synth.make_molecules(chunk_mols, 0, seed=seed + c, dist=dist_name, edge_features=edge_features)
              for c in range(num_chunks)]
    per_mol = np.concatenate([np.add.reduceat(np.diff(c.row_ptr).astype(np.int64), c.atom_ptr[:-1].astype(np.int64))
                              for c in chunks])
    mine = shard_by_edges(per_mol, world)[rank]                       # global ids, ascending
    parts, gids = [], []
    for c, chunk in enumerate(chunks):
        lo, hi = c * chunk_mols, (c + 1) * chunk_mols
        sel = mine[(mine >= lo) & (mine < hi)]
        if sel.size:
            parts.append(synth.select(chunk, sel - lo) if sel.size < chunk_mols else chunk)
            gids.append(sel)
    local = synth.concat(parts)
    gids = np.concatenate(gids)
    nmb = max(1, -(-local.num_mols // micro_mols))
    bounds = np.linspace(0, local.num_mols, nmb + 1).astype(np.int64)
    micro = []
    for i in range(nmb):
        ids = np.arange(bounds[i], bounds[i + 1])
        mb = local if nmb == 1 else synth.select(local, ids)
        micro.append((mb, synth.hashed_atom_keys(gids[ids], mb.n_atoms)))
    info = {"global_mols": int(per_mol.shape[0]), "global_edges": int(per_mol.sum()),
            "local_mols": int(local.num_mols), "local_edges": int(local.num_edges)}
    return micro, info
