"""AdjMsgAgg: out_i = sum_j adj_ij * m_ij  -> the destination-sorted segmented sum kernel.

Reference: mpnn_functions/message_aggregators/adjacent_message_agg.py:4-18.  adj values act as
multipliers (any float), so they ride along as per-edge weights of mpnn_segsum_f32; pairs with
adj == 0 contribute nothing, hence no non-member correction here.
"""
from torch import nn

from mpnn_amd import ops
from mpnn_amd.messages import EdgeMessages
from ._common import adjacency_multiplier, dense_rows


class AdjMsgAgg(nn.Module):
    def __init__(self, adj_dim, attn_act=None):
        super().__init__()

    def forward(self, messages, adj):
        if isinstance(messages, EdgeMessages):
            g = messages.graph
            w = adjacency_multiplier(messages, adj)
            if messages._values is None:          # lazy messages: message + sum as one autograd node
                A, gate = messages.recipe
                if isinstance(gate, ops.LazyAttGate):     # AttEdgeNetwork: the gate is formed inside the fused kernel where it can
                    return g.node_unview(ops.gated_message_aggregate(messages.h, A, gate, g, w))
                return g.node_unview(ops.message_aggregate(messages.h, A, g, w, gate))
            return g.node_unview(ops.segsum(messages.values, g.row_ptr, w))
        rows, row_ptr, (B, N) = dense_rows(messages, adj)
        return ops.segsum(rows, row_ptr, adj.reshape(-1).contiguous().float()).view(B, N, -1)
