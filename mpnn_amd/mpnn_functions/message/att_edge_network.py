"""AttEdgeNetwork: feature-wise gate on the source atom, then the edge-network product.

Reference: mpnn_functions/message/att_edge_network.py:6-31.
    gate_ij = attn_act(attn([h_i, e_ij]))      softmax over the FEATURE axis by default
    m_ij    = A(e_ij) . (gate_ij * h_j)
The Linear over the concatenation splits into an atom part (once per atom) and a bond part (once
per distinct bond-feature row): z_ij = W_h h_i + W_e e_ij + b.  The gated product is the `gate`
argument of mpnn_edge_message_f32.  Always returns per-pair messages (as the reference does).
"""
import torch
from torch import nn

from mpnn_amd import ops
from mpnn_amd.messages import EdgeMessages
from .edge_network import EdgeNetwork


class AttEdgeNetwork(EdgeNetwork):
    def __init__(self, node_features, edge_features, message_features, activation_fn=None, attn_act=None):
        super().__init__(node_features, edge_features, message_features, activation_fn)
        self.attn = nn.Linear(self.nf + self.ef, self.nf)
        self.attn_act = attn_act if attn_act is not None else nn.Softmax(dim=-1)
        self.pairwise = True

    def forward(self, afm, bfm, reuse_graph_tensors=False):
        if not reuse_graph_tensors or self.edge_embed is None:
            self._precompute_edge_embed(bfm)
        emb = self.edge_embed
        g = emb.graph
        h = g.node_view(afm)
        W_h, W_e = self.attn.weight[:, :self.nf], self.attn.weight[:, self.nf:]
        z_atom = ops.tall_linear(h, W_h, self.attn.bias)      # (V, nf): destination-atom part
        if isinstance(self.attn_act, nn.Softmax) and self.attn_act.dim in (-1, 1) and self.nf % 4 == 0 and self.nf <= 256:
            # default activation: one streaming kernel gathers both parts of the logits and applies the softmax
            # over the feature axis (mpnn_att_gate_f32); the bond part is a (K, nf) table, one row per distinct
            # bond-feature row
            # -- lazily: followed by the sum aggregator at hidden 128 the gate is never written out (ops.LazyAttGate)
            gate = ops.LazyAttGate(z_atom, g.type_feat @ W_e.t(), g)
        else:
            # any other activation: the atom part is broadcast along CSR rows (backward = the aggregator kernel),
            # the bond part is a thin GEMM on the edge features (backward = a GEMM) -- no index_put backward
            gate = self.attn_act(ops.expand_rows(z_atom, g) + g.edge_features @ W_e.t())
        return EdgeMessages(None, g, h, emb.A0, row_gate=lambda: self.attn_act(z_atom), recipe=(emb.A, gate))
