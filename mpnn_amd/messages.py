"""Sparse stand-in for the reference's per-pair message tensor (B,N,N,mf).

The reference materialises a message for EVERY ordered pair, non-bonded and padded pairs
included (mpnn_functions/message/edge_network.py:34-40); aggregators then weigh pairs
(message_aggregators/*.py).  Here only the E member pairs carry a row; every other pair of a
molecule has the same matrix A0 = edge_map(0), so its message is A0 . (gate_i0 * h_j) and the
sum over those pairs collapses to one per-atom term (`nonedge_sum`).
"""
import torch

from . import ops


class EdgeMessages:
    def __init__(self, values, graph, h, A0=None, row_gate=None, recipe=None):
        self._values = values     # (E, mf) message of each member pair, destination-sorted (None = lazy)
        self.graph = graph
        self.h = h                # (V, nf) node features the messages were computed from
        self.A0 = A0              # (mf, nf) matrix of a zero bond-feature row, or None (== 0)
        self._row_gate = row_gate  # (V, nf) AttEdgeNetwork gate of atom i towards a zero-feature pair, or a thunk
        self.recipe = recipe      # (A, gate): how to compute `values`; lets AdjMsgAgg run message+sum as one
                                  # autograd node (ops.message_aggregate) whose backward skips the (E, mf) gradient

    @property
    def row_gate(self):
        """Evaluated on first use: only the non-member corrections and to_dense() need it."""
        if callable(self._row_gate):
            self._row_gate = self._row_gate()
        return self._row_gate

    @property
    def values(self):
        if self._values is None:
            A, gate = self.recipe
            if isinstance(gate, ops.LazyAttGate):
                gate = gate.materialise()
            self._values = ops.edge_message(self.h, A, self.graph, gate=gate)
        return self._values

    @property
    def num_features(self):
        return int(self.recipe[0].shape[1]) if self._values is None else int(self._values.shape[-1])

    @property
    def shape(self):
        g = self.graph
        if g.dense_shape is None:
            raise RuntimeError("compact batches have no dense (B,N,N,mf) shape")
        B, N = g.dense_shape
        return (B, N, N, self.num_features)

    def nonedge_sum(self):
        """(V, mf): sum over the NON-member pairs (i, j) of molecule(i) of their message.
        = A0 . (gate_i0 * (S_mol(i) - sum_{e in row i} h_src(e)))   (padded atoms have h = 0)."""
        g = self.graph
        mf = self.num_features
        if self.A0 is None:
            return torch.zeros(g.num_nodes, mf, device=self.h.device)
        rest = ops.molecule_broadcast(ops.molecule_sum(self.h, g), g) - ops.neighbour_sum(self.h, g)
        if self.row_gate is not None:
            rest = rest * self.row_gate
        return rest @ self.A0.t()

    def to_dense(self):
        """The reference's (B,N,N,mf) tensor (compat / debugging only; O(B N^2 mf) memory)."""
        g = self.graph
        B, N = g.dense_shape
        mf = self.values.shape[-1]
        h = self.h.view(B, N, -1)
        if self.A0 is None:
            dense = torch.zeros(B, N, N, mf, device=h.device)
        elif self.row_gate is None:
            dense = (h @ self.A0.t()).unsqueeze(1).expand(B, N, N, mf).clone()
        else:
            x = self.row_gate.view(B, N, 1, -1) * h.view(B, 1, N, -1)
            dense = x @ self.A0.t()
        dst = g.edge_dst.to(torch.int64)
        src = g.col_idx.to(torch.int64)
        flat = dst * N + (src - (dst // N) * N)
        dense = dense.reshape(B * N * N, mf).index_put((flat,), self.values)
        return dense.view(B, N, N, mf)
