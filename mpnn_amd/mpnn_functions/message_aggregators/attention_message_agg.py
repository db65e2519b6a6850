"""AttMsgAgg: out_i = sum_j act(w * adj_ij + b) * m_ij over ALL pairs j of the padded row.

Reference: mpnn_functions/message_aggregators/attention_message_agg.py:5-24.  The Linear acts on
adj.unsqueeze(-1), so it only works for adj_dim == 1; with the default Softmax(dim=-1) over that
size-1 axis every weight is exactly 1 (plain all-pairs sum).  Non-member pairs all share the weight
act(b), so they enter through EdgeMessages.nonedge_sum().
"""
import torch
from torch import nn

from mpnn_amd import ops
from mpnn_amd.messages import EdgeMessages
from ._common import dense_rows, edge_adjacency


class AttMsgAgg(nn.Module):
    def __init__(self, adj_dim, attn_act=None):
        super().__init__()
        self.adj_dim = adj_dim
        self.att = nn.Sequential(
            nn.Linear(adj_dim, 1),
            attn_act if attn_act is not None else nn.Softmax(dim=-1),
        )

    def forward(self, messages, adj):
        if isinstance(messages, EdgeMessages):
            g = messages.graph
            if isinstance(self.att[1], nn.Softmax) and self.att[1].dim == -1 and messages._values is None:
                # default activation: a softmax over the size-1 last axis is identically 1 (and its gradient towards the
                # Linear is identically 0), so the member pairs are a plain sum -- message + sum as ONE autograd node,
                # exactly as AdjMsgAgg does -- and every non-member pair of the padded row joins with weight 1
                A, gate = messages.recipe
                if isinstance(gate, ops.LazyAttGate):
                    gate = gate.materialise()
                out = ops.message_aggregate(messages.h, A, g, None, gate) + messages.nonedge_sum()
                return g.node_unview(out)
            w = self.att(edge_adjacency(messages, adj).unsqueeze(-1)).squeeze(-1).contiguous()
            w0 = self.att(torch.zeros(1, 1, device=w.device)).reshape(())
            out = ops.segsum(messages.values, g.row_ptr, w) + w0 * messages.nonedge_sum()
            return g.node_unview(out)
        rows, row_ptr, (B, N) = dense_rows(messages, adj)
        w = self.att(adj.unsqueeze(-1)).reshape(-1).contiguous()
        return ops.segsum(rows, row_ptr, w).view(B, N, -1)
