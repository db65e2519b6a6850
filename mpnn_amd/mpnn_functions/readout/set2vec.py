"""Set2Vec readout (Vinyals et al. order-invariant read) as the reference wires it.

Reference: mpnn_functions/readout/set2vec.py:12-151 (LSTMCellHidden :12-75, Set2Vec :78-151).  Kept quirks:
  * `output_dim` and `dropout` are accepted and ignored; the result is cat[m, read], (B, 4*node_features);
  * the attention softmax runs over dim 0 of the FLATTENED (B*N, 1) energies (:139), i.e. across every atom of
    the whole batch, not per molecule; padded atoms are pushed to -1e8 first, so they weigh exactly 0;
  * the "dot" inner product adds a (B*N,1) mask to (B,N) energies (:136-137), which only broadcasts when that
    happens to be legal; it is supported here without a mask, as in the reference.
The per-step LSTM and the query projection are (B, .) library GEMMs; the per-atom part runs either on the dense
(B,N,.) layout exactly as written, or, for a compact batch, on (V,.) rows with the per-molecule broadcast /
sum done by the aggregator kernels (graph_ptr as row_ptr).
"""
import math

import torch
from torch import nn

from mpnn_amd import ops
from mpnn_amd.graph import MolGraph

_BIG_NEGATIVE = -1e8


class LSTMCellHidden(nn.Module):
    """set2vec.py:12-75: an LSTM cell driven by the previous hidden state only (no input)."""

    def __init__(self, hidden_dim, cell_dim, bias=True):
        super().__init__()
        self.hd, self.cd, self.bias = hidden_dim, cell_dim, bias
        for gate in ("i", "f", "g", "o"):
            self.register_parameter("w_h" + gate, nn.Parameter(torch.zeros(self.hd, self.cd)))
        for gate in ("i", "f", "g", "o"):
            self.register_parameter("b_h" + gate, nn.Parameter(torch.zeros(1, self.cd)))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.hd)
        for gate in ("i", "f", "g", "o"):
            getattr(self, "w_h" + gate).data.uniform_(-stdv, stdv)

    def forward(self, hprev, cprev):
        i = torch.sigmoid(hprev.matmul(self.w_hi) + self.b_hi)
        f = torch.sigmoid(hprev.matmul(self.w_hf) + self.b_hf)
        g = torch.tanh(hprev.matmul(self.w_hg) + self.b_hg)
        o = torch.sigmoid(hprev.matmul(self.w_ho) + self.b_ho)
        cprime = f * cprev + i * g
        return o * torch.tanh(cprime), cprime


class Set2Vec(nn.Module):
    def __init__(self, node_features, output_dim, time_steps=100, inner_prod="default", activation_fn=None,
                 attn_act=None, dropout=0):
        super().__init__()
        self.nf = 2 * node_features
        self.steps = time_steps
        self.q_attn = nn.Linear(self.nf, self.nf, bias=False)
        if inner_prod == "default":
            self.ip = True
            self.e_attn = nn.Linear(self.nf, 1, bias=False)
        elif inner_prod == "dot":
            self.ip = False
        else:
            raise ValueError("Invalid inner_prod type: {}".format(inner_prod))
        self.add_module("lstmcell", LSTMCellHidden(self.nf * 2, self.nf))

    def forward(self, input_set, mask=None, mprev=None, cprev=None, graph=None):
        compact = isinstance(graph, MolGraph) and input_set.dim() == 2
        batch_size = graph.num_graphs if compact else input_set.shape[0]
        zeros = input_set.new_zeros(batch_size, self.nf)
        if mprev is None:
            mprev = zeros
        mprev = torch.cat([mprev, zeros], dim=1)
        if cprev is None:
            cprev = zeros
        if compact and not self.ip:
            raise ValueError("Set2Vec(inner_prod='dot') is defined on the dense (B,N,.) layout only")
        penalty = (1 - mask) * _BIG_NEGATIVE if mask is not None else None
        m = mprev
        for _ in range(self.steps):
            m, c = self.lstmcell(mprev, cprev)
            query = self.q_attn(m)                                        # (B, nf)
            if compact:
                q_rows = ops.molecule_broadcast(query, graph)             # (V, nf)
                energies = self.e_attn(torch.tanh(q_rows + input_set))    # (V, 1)
                if penalty is not None:
                    energies = energies + penalty.view(-1, 1)
                att = torch.softmax(energies, dim=0)                      # across the whole batch (:139)
                read = ops.molecule_sum(att * input_set, graph)           # (B, nf)
            else:
                if self.ip:
                    energies = self.e_attn(torch.tanh(query.unsqueeze(1) + input_set).view(-1, self.nf))
                else:
                    energies = input_set.matmul(query.view(-1, self.nf, 1)).view(batch_size, -1)
                if penalty is not None:
                    energies = energies + penalty.view(-1, 1)
                att = torch.softmax(energies, dim=0).view(batch_size, -1, 1)
                read = att.mul(input_set).sum(dim=1)
            m = torch.cat([m, read], dim=1)
            mprev, cprev = m, c
        return m
