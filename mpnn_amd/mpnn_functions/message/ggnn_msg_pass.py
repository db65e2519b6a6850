"""GGNNMsgPass: integer bond types index a learned table of (mf x nf) matrices.

Reference: mpnn_functions/message/ggnn_msg_pass.py:4-31.  Type 0 means "no bond" and maps to an
exact zero matrix, so the fused all-pairs sum of the reference is a plain sum over real edges:
the table itself is the `A` operand of mpnn_edge_message_f32 and the tower disappears.
`bfm` is the integer (B,N,N) bond-type tensor, or a MolGraph whose edge types are 0-based rows
of `adj_w`.
"""
import torch
from torch import nn

from mpnn_amd import ops
from mpnn_amd.graph import MolGraph, _i32


class GGNNMsgPass(nn.Module):
    def __init__(self, node_features, edge_features, message_features):
        super().__init__()
        self.nf = node_features
        self.ef = edge_features
        self.mf = message_features
        self.adj_w = nn.Parameter(torch.empty(self.ef, self.mf, self.nf))
        self.message_bias = nn.Parameter(torch.zeros(self.mf))
        self.zeros = nn.Parameter(torch.zeros(1, self.mf, self.nf), requires_grad=False)
        self.edge_embed = None
        self.init_weights()

    def init_weights(self):
        nn.init.kaiming_uniform_(self.adj_w, nonlinearity='relu')

    def _precompute_edge_embed(self, bfm):
        if isinstance(bfm, MolGraph):
            self.edge_embed = bfm
            return
        g = MolGraph.from_dense(bfm.float(), None)            # pairs with a non-zero type
        B, N = g.dense_shape
        dst = g.edge_dst.to(torch.int64)
        flat = dst * N + (g.col_idx.to(torch.int64) - (dst // N) * N)
        g.edge_type = _i32(bfm.reshape(-1)[flat] - 1)
        g.type_feat = torch.eye(self.ef, device=bfm.device)
        g.num_types = self.ef
        g.edge_weight = None
        self.edge_embed = g

    def forward(self, afm, bfm, reuse_graph_tensors=False):
        if not reuse_graph_tensors or self.edge_embed is None:
            self._precompute_edge_embed(bfm)
        g = self.edge_embed
        h = g.node_view(afm)
        msg = ops.edge_message(h, self.adj_w, g)
        return g.node_unview(ops.segsum(msg, g.row_ptr) + self.message_bias)
