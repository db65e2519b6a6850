"""WAdjMsgAgg: out_i = sum_j softmax_j(adj_i.)_j * m_ij, softmax over the PADDED row.

Reference: mpnn_functions/message_aggregators/weighted_adjacent_message_agg.py:6-20.  Non-bonded
and padded pairs weigh exp(0)/Z_i each, so with d_i member pairs in a row of padded length N:
    Z_i  = sum_{e in row i} exp(adj_e) + (N - d_i)
    out_i = (sum_e exp(adj_e) m_e + nonmember_sum_i) / Z_i
"""
import torch
from torch import nn

from mpnn_amd import ops
from mpnn_amd.messages import EdgeMessages
from ._common import dense_rows, edge_adjacency


class WAdjMsgAgg(nn.Module):
    def __init__(self, adj_dim, attn_act=None):
        super().__init__()

    def forward(self, messages, adj):
        if isinstance(messages, EdgeMessages):
            g = messages.graph
            ex = torch.exp(edge_adjacency(messages, adj))
            deg = (g.row_ptr[1:] - g.row_ptr[:-1]).float()
            Z = ops.segsum(ex.unsqueeze(-1).contiguous(), g.row_ptr).squeeze(-1) + (g.pad_size - deg)
            num = ops.segsum(messages.values, g.row_ptr, ex) + messages.nonedge_sum()
            return g.node_unview(num / Z.unsqueeze(-1))
        rows, row_ptr, (B, N) = dense_rows(messages, adj)
        w = torch.softmax(adj, dim=-1).reshape(-1).contiguous()
        return ops.segsum(rows, row_ptr, w).view(B, N, -1)
