"""mpnn_amd.collate against the contract of the reference's dense collate (pre_process/data_loader.py:50-70): the same
list of per-molecule objects in; padded to the largest molecule, the sparse batch must carry exactly the dense batch."""
import numpy as np
import torch

from mpnn_amd import synth
from mpnn_amd.collate import collate_sparse, to_dense


class G2D:
    """What mol_graph.Graph2D instances look like to the collate: numpy afm / nafm / bfm / adj and a label."""

    def __init__(self, afm, nafm, bfm, adj, label):
        self.afm, self.nafm, self.bfm, self.adj, self.label = afm, nafm, bfm, adj, label


def fake_graphs(n_mols, seed, af=19, naf=3, ef=7):
    mb = synth.make_molecules(n_mols, af + naf, seed=seed, dist="lipo", edge_features=ef, lipo_features=True)
    rng = np.random.default_rng(seed)
    out = []
    for g in range(mb.num_mols):
        d = synth.to_dense(synth.select(mb, [g]), numeric_tail=naf)
        out.append(G2D(d["afm"][0], d["nafm"][0], d["bfm"][0], d["adj"][0], float(rng.normal())))
    return out


def dense_contract(graphs):
    """The padded batch as the reference's collate lays it out (embed_arr / create_mask, data_loader.py:12-21)."""
    N = max(g.afm.shape[0] for g in graphs)

    def pad(a, shape):
        out = np.zeros(shape, np.float32)
        out[tuple(slice(0, s) for s in a.shape)] = a
        return out
    return {
        "afm": np.stack([pad(g.afm, (N, g.afm.shape[1])) for g in graphs]),
        "nafm": np.stack([pad(g.nafm, (N, g.nafm.shape[1])) for g in graphs]),
        "bfm": np.stack([pad(g.bfm, (N, N, g.bfm.shape[2])) for g in graphs]),
        "adj": np.stack([pad(g.adj, (N, N)) for g in graphs]),
        "mask": np.stack([pad(np.ones((g.afm.shape[0], 1), np.float32), (N, 1)) for g in graphs]),
        "labels": np.array([g.label for g in graphs]),
    }


def test_sparse_batch_unpacks_to_the_dense_contract_bit_for_bit():
    graphs = fake_graphs(16, seed=3)
    want = dense_contract(graphs)
    got = to_dense(collate_sparse(graphs, torch.device("cpu")))
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(got[k].numpy(), want[k]), k


def test_csr_is_the_nonzero_order_of_the_padded_adjacency():
    graphs = fake_graphs(9, seed=4)
    want = dense_contract(graphs)
    b = collate_sparse(graphs, torch.device("cpu"))
    g = b["graph"]
    N = want["adj"].shape[1]
    bb, ii, jj = np.nonzero(want["adj"])
    gp = g.graph_ptr.numpy().astype(np.int64)
    assert np.array_equal(g.edge_dst.numpy(), gp[bb] + ii)            # compact ids, (molecule, dst, src) order
    assert np.array_equal(g.col_idx.numpy(), gp[bb] + jj)
    assert np.array_equal(g.type_feat.numpy()[g.edge_type.numpy()], want["bfm"][bb, ii, jj])
    assert g.num_types <= 7 and g.num_graphs == 9 and g.num_nodes == int(b["n_atoms"].sum())
    assert float(g.pad_size.min()) == float(g.pad_size.max()) == N     # padded-row operators see the dense N


def test_bond_features_without_adjacency_still_make_an_edge_and_empty_batches_collate():
    a = G2D(np.eye(3, 4, dtype=np.float32), np.zeros((3, 2), np.float32), np.zeros((3, 3, 2), np.float32),
            np.zeros((3, 3), np.float32), 1.0)
    a.bfm[0, 2, 1] = a.bfm[2, 0, 1] = 0.5                             # a pair the adjacency does not list
    b = collate_sparse([a], torch.device("cpu"))
    g = b["graph"]
    assert g.num_edges == 2 and g.edge_weight.tolist() == [0.0, 0.0]
    assert g.col_idx.tolist() == [2, 0] and g.edge_dst.tolist() == [0, 2]
    lone = G2D(np.ones((1, 4), np.float32), np.zeros((1, 2), np.float32), np.zeros((1, 1, 2), np.float32),
               np.zeros((1, 1), np.float32), 0.0)
    b = collate_sparse([lone, lone], torch.device("cpu"))
    assert b["graph"].num_edges == 0 and b["graph"].num_nodes == 2 and b["afm"].shape == (2, 4)
