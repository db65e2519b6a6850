"""Synthetic molecular graphs (numpy only, no torch, no GPU).

Shapes follow SURVEY.md §8(d): per-molecule atom count, a random bond tree plus
0..2 ring closures (directed edges E_g = 2*(n_g - 1 + r_g)), one-hot bond-type
edge features, uniform atom features.  The generator is vectorised *across*
molecules (one numpy step per atom position), so 100k..1M molecules build in
seconds.

Two output forms:

* :class:`MolBatch` -- the sparse form the HIP path consumes: atoms numbered
  compactly over the whole batch, directed edges sorted by (molecule, dst, src),
  which is exactly the order ``adj.nonzero()`` yields on the dense form.
* :func:`to_dense` -- the padded dict batch the reference's collate emits
  (keys afm/nafm/bfm/adj/mask, reference: pre_process/data_loader.py:50-70).
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np

BOND_TYPE_P = (0.55, 0.30, 0.13, 0.02)  # single / double / aromatic / triple


@dataclass
class MolBatch:
    n_atoms: np.ndarray      # (G,)   int32 atoms per molecule
    atom_ptr: np.ndarray     # (G+1,) int32 first atom of each molecule (compact numbering)
    row_ptr: np.ndarray      # (V+1,) int32 CSR by destination atom
    col_idx: np.ndarray      # (E,)   int32 source atom of each directed edge
    bond_type: np.ndarray    # (E,)   int32 in [0, K) -- row of `type_feat`
    type_feat: np.ndarray    # (K,ef) float32 distinct edge-feature rows
    edge_feat: Optional[np.ndarray]  # (E,ef) float32, only for the continuous variant
    atom_feat: np.ndarray    # (V,nf) float32

    @property
    def num_mols(self):
        return int(self.n_atoms.shape[0])

    @property
    def num_atoms(self):
        return int(self.atom_ptr[-1])

    @property
    def num_edges(self):
        return int(self.col_idx.shape[0])

    def edge_features(self):
        if self.edge_feat is not None:
            return self.edge_feat
        return self.type_feat[self.bond_type]


def _atom_counts(rng, G, dist):
    if dist == "drug":          # C2/C3/C4: round(N(30,7)) clipped to [5,50]
        n = np.rint(rng.normal(30.0, 7.0, size=G)).astype(np.int64)
        return np.clip(n, 5, 50)
    if dist == "skewed":        # C5: uniform [10,200]
        return rng.integers(10, 201, size=G)
    if dist == "lipo":          # C1: uniform [10,50]
        return rng.integers(10, 51, size=G)
    raise ValueError("unknown size distribution %r" % (dist,))


def _random_trees(rng, n, preferential, max_degree):
    """parent[g, i] for i in 1..n_g-1 (parent < i); -1 elsewhere."""
    G = n.shape[0]
    nmax = int(n.max())
    parent = np.full((G, nmax), -1, dtype=np.int32)
    if preferential:
        # endpoint list per molecule; drawing a uniform endpoint == degree-proportional draw
        ends = np.zeros((G, 2 * nmax), dtype=np.int32)
        for i in range(1, nmax):
            live = np.nonzero(n > i)[0]
            if live.size == 0:
                break
            if i == 1:
                p = np.zeros(live.size, dtype=np.int32)
            else:
                pick = rng.integers(0, 2 * (i - 1), size=live.size)
                p = ends[live, pick]
            parent[live, i] = p
            ends[live, 2 * (i - 1)] = p
            ends[live, 2 * (i - 1) + 1] = i
        return parent
    deg = np.zeros((G, nmax), dtype=np.int8)
    for i in range(1, nmax):
        live = np.nonzero(n > i)[0]
        if live.size == 0:
            break
        p = rng.integers(0, i, size=live.size).astype(np.int32)
        if max_degree is not None:
            for _ in range(64):
                bad = deg[live, p] >= max_degree
                if not bad.any():
                    break
                p[bad] = rng.integers(0, i, size=int(bad.sum()))
            bad = deg[live, p] >= max_degree
            if bad.any():
                # deterministic fallback: first atom that still has a free valence
                free = deg[live[bad], :i] < max_degree
                p[bad] = free.argmax(axis=1)
        parent[live, i] = p
        deg[live, p] += 1
        deg[live, i] += 1
    return parent


def _ring_closures(rng, n, parent, want):
    """Up to two extra bonds per molecule between non-bonded distinct atoms."""
    G = n.shape[0]
    ring = np.full((G, 2, 2), -1, dtype=np.int32)
    for slot in range(2):
        todo = np.nonzero(want > slot)[0]
        for _ in range(8):
            if todo.size == 0:
                break
            u = (rng.random(todo.size) * n[todo]).astype(np.int32)
            v = (rng.random(todo.size) * n[todo]).astype(np.int32)
            lo, hi = np.minimum(u, v), np.maximum(u, v)
            ok = lo != hi
            ok &= parent[todo, hi] != lo          # not a tree bond (parent index < child index)
            if slot == 1:
                ok &= ~((ring[todo, 0, 0] == lo) & (ring[todo, 0, 1] == hi))
            ring[todo[ok], slot, 0] = lo[ok]
            ring[todo[ok], slot, 1] = hi[ok]
            todo = todo[~ok]
    return ring


def make_molecules(num_mols, node_features, seed=317, dist="drug", edge_features=4,
                   continuous=False, preferential=None, lipo_features=False, atom_features=True):
    """Build a :class:`MolBatch` of `num_mols` synthetic molecules.  atom_features=False: graph structure only (`atom_feat`
    is a (V, 0) array; the caller makes the features on the device, `hashed_features`) -- at hidden 256 the host-side
    random features are 5 GB and most of the generation time."""
    rng = np.random.default_rng(seed)
    G = int(num_mols)
    n = _atom_counts(rng, G, dist).astype(np.int64)
    if preferential is None:
        preferential = dist == "skewed"
    parent = _random_trees(rng, n, preferential, None if preferential else 4)
    want = rng.choice(3, size=G, p=(0.25, 0.5, 0.25))
    want = np.minimum(want, np.maximum(n - 3, 0))
    ring = _ring_closures(rng, n, parent, want)

    atom_ptr = np.zeros(G + 1, dtype=np.int64)
    np.cumsum(n, out=atom_ptr[1:])
    V = int(atom_ptr[-1])

    # undirected bond list (mol, a, b) in local atom numbering
    gi, ci = np.nonzero(parent >= 0)
    b_mol = [gi]
    b_a = [ci.astype(np.int64)]
    b_b = [parent[gi, ci].astype(np.int64)]
    for slot in range(2):
        g2 = np.nonzero(ring[:, slot, 0] >= 0)[0]
        b_mol.append(g2)
        b_a.append(ring[g2, slot, 0].astype(np.int64))
        b_b.append(ring[g2, slot, 1].astype(np.int64))
    b_mol = np.concatenate(b_mol)
    b_a = np.concatenate(b_a)
    b_b = np.concatenate(b_b)
    nb = b_mol.shape[0]

    K = int(edge_features)
    if continuous:
        bond_feat = rng.random((nb, K), dtype=np.float32)
        btype = None
    else:
        p = np.asarray(BOND_TYPE_P[:K], dtype=np.float64) if K <= 4 else np.full(K, 1.0 / K)
        p = p / p.sum()
        btype = rng.choice(K, size=nb, p=p).astype(np.int32)

    # both directions, then sort by (dst, src) on global atom ids == (mol, dst, src)
    base = atom_ptr[b_mol]
    dst = np.concatenate([base + b_a, base + b_b])
    src = np.concatenate([base + b_b, base + b_a])
    order = np.lexsort((src, dst))
    dst = dst[order]
    src = src[order]
    E = dst.shape[0]
    row_ptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(np.bincount(dst, minlength=V), out=row_ptr[1:])

    if continuous:
        edge_feat = np.concatenate([bond_feat, bond_feat])[order]
        type_feat = edge_feat  # every edge is its own "type"; callers dedupe if they want
        bond_type = np.arange(E, dtype=np.int32)
    else:
        edge_feat = None
        bond_type = np.concatenate([btype, btype])[order]
        type_feat = np.eye(K, dtype=np.float32)

    nf = int(node_features)
    if not atom_features:
        atom_feat = np.zeros((V, 0), dtype=np.float32)
    elif lipo_features:
        # C1 shape: 19 one-hot columns + 3 numeric columns in [0,1)
        onehot = max(nf - 3, 1)
        atom_feat = np.zeros((V, nf), dtype=np.float32)
        atom_feat[np.arange(V), rng.integers(0, onehot, size=V)] = 1.0
        atom_feat[:, onehot:] = rng.random((V, nf - onehot), dtype=np.float32)
    else:
        atom_feat = (rng.random((V, nf), dtype=np.float32) * 2.0 - 1.0).astype(np.float32)

    return MolBatch(
        n_atoms=n.astype(np.int32),
        atom_ptr=atom_ptr.astype(np.int32),
        row_ptr=row_ptr.astype(np.int32),
        col_idx=src.astype(np.int32),
        bond_type=bond_type.astype(np.int32),
        type_feat=type_feat,
        edge_feat=edge_feat,
        atom_feat=atom_feat,
    )


def select(batch, mol_ids):
    """Sub-batch holding the molecules `mol_ids` (kept in the given order)."""
    mol_ids = np.asarray(mol_ids, dtype=np.int64)
    n = batch.n_atoms[mol_ids].astype(np.int64)
    new_ptr = np.zeros(mol_ids.shape[0] + 1, dtype=np.int64)
    np.cumsum(n, out=new_ptr[1:])
    V = int(new_ptr[-1])
    old_start = batch.atom_ptr[mol_ids].astype(np.int64)
    # old atom id of every new atom
    rep = np.repeat(np.arange(mol_ids.shape[0]), n)
    old_atom = old_start[rep] + (np.arange(V) - new_ptr[rep])
    remap = np.full(batch.num_atoms, -1, dtype=np.int64)
    remap[old_atom] = np.arange(V)
    deg = (batch.row_ptr[old_atom + 1] - batch.row_ptr[old_atom]).astype(np.int64)
    row_ptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(deg, out=row_ptr[1:])
    E = int(row_ptr[-1])
    rep_e = np.repeat(np.arange(V), deg)
    old_edge = batch.row_ptr[old_atom][rep_e].astype(np.int64) + (np.arange(E) - row_ptr[rep_e])
    col = remap[batch.col_idx[old_edge]]
    assert (col >= 0).all()
    return MolBatch(
        n_atoms=n.astype(np.int32),
        atom_ptr=new_ptr.astype(np.int32),
        row_ptr=row_ptr.astype(np.int32),
        col_idx=col.astype(np.int32),
        bond_type=(batch.bond_type[old_edge] if batch.edge_feat is None
                   else np.arange(E, dtype=np.int32)),
        type_feat=(batch.type_feat if batch.edge_feat is None else batch.edge_feat[old_edge]),
        edge_feat=(None if batch.edge_feat is None else batch.edge_feat[old_edge]),
        atom_feat=batch.atom_feat[old_atom],
    )


def to_dense(batch, numeric_tail=0):
    """Padded dict batch in the reference's wire format (numpy arrays).

    `numeric_tail` > 0 splits the last columns of the atom features off as 'nafm'
    (the lipo wrapper normalises them separately, reference:
    models/graph_norm_wrapper.py:12-13).
    """
    G = batch.num_mols
    N = int(batch.n_atoms.max())
    nf = batch.atom_feat.shape[1]
    ef = batch.type_feat.shape[1]
    afm = np.zeros((G, N, nf), dtype=np.float32)
    bfm = np.zeros((G, N, N, ef), dtype=np.float32)
    adj = np.zeros((G, N, N), dtype=np.float32)
    mask = np.zeros((G, N, 1), dtype=np.float32)
    V = batch.num_atoms
    mol_of_atom = np.repeat(np.arange(G), batch.n_atoms)
    local = np.arange(V) - batch.atom_ptr[mol_of_atom]
    afm[mol_of_atom, local] = batch.atom_feat
    mask[mol_of_atom, local, 0] = 1.0
    deg = np.diff(batch.row_ptr)
    dst = np.repeat(np.arange(V), deg)
    src = batch.col_idx
    g = mol_of_atom[dst]
    adj[g, local[dst], local[src]] = 1.0
    bfm[g, local[dst], local[src]] = batch.edge_features()
    out = {"afm": afm, "bfm": bfm, "adj": adj, "mask": mask}
    if numeric_tail:
        out["afm"] = afm[..., : nf - numeric_tail].copy()
        out["nafm"] = afm[..., nf - numeric_tail:].copy()
    return out


def concat(batches):
    """One MolBatch holding the molecules of `batches` in order (discrete bond types only: the batches must share
    their `type_feat` table)."""
    if len(batches) == 1:
        return batches[0]
    if any(b.edge_feat is not None for b in batches):
        raise ValueError("concat: continuous bond features are not supported")
    n = np.concatenate([b.n_atoms for b in batches]).astype(np.int64)
    atom_ptr = np.zeros(n.shape[0] + 1, dtype=np.int64)
    np.cumsum(n, out=atom_ptr[1:])
    a_off = np.cumsum([0] + [b.num_atoms for b in batches])
    e_off = np.cumsum([0] + [b.num_edges for b in batches])
    row_ptr = np.concatenate([batches[0].row_ptr[:1].astype(np.int64)] +
                             [b.row_ptr[1:].astype(np.int64) + e_off[i] for i, b in enumerate(batches)])
    col = np.concatenate([b.col_idx.astype(np.int64) + a_off[i] for i, b in enumerate(batches)])
    return MolBatch(n_atoms=n.astype(np.int32), atom_ptr=atom_ptr.astype(np.int32), row_ptr=row_ptr.astype(np.int32),
                    col_idx=col.astype(np.int32), bond_type=np.concatenate([b.bond_type for b in batches]),
                    type_feat=batches[0].type_feat, edge_feat=None,
                    atom_feat=np.concatenate([b.atom_feat for b in batches], axis=0))


def hashed_atom_keys(mol_gid, n_atoms):
    """int64 key of every atom of a batch = (global molecule id, index inside the molecule): a feature generator keyed
    on it gives a molecule the same features whichever rank / micro-batch it lands in."""
    mol_gid = np.asarray(mol_gid, dtype=np.int64)
    n = np.asarray(n_atoms, dtype=np.int64)
    ptr = np.zeros(n.shape[0] + 1, dtype=np.int64)
    np.cumsum(n, out=ptr[1:])
    rep = np.repeat(np.arange(n.shape[0]), n)
    return mol_gid[rep] * 256 + (np.arange(int(ptr[-1])) - ptr[rep])


def hashed_features(keys, node_features, device=None):
    """(V, nf) uniform(-1, 1) fp32 features as a pure function of (atom key, column), computed with torch integer ops
    on `device` (a 32-bit mix of key * nf + column): 1 M molecules x 128 features never exist on the host."""
    import torch
    k = torch.as_tensor(keys, dtype=torch.int64, device=device)
    x = (k.unsqueeze(1) * int(node_features) + torch.arange(int(node_features), device=k.device, dtype=torch.int64))
    m = 0xFFFFFFFF
    x = (x ^ (x >> 16)) & m
    x = (x * 0x45D9F3B) & m
    x = (x ^ (x >> 16)) & m
    x = (x * 0x45D9F3B) & m
    x = (x ^ (x >> 16)) & m
    return (x.to(torch.float32) * (2.0 / 4294967296.0) - 1.0).contiguous()
