"""GraphLevelOutput readout: softmax-gated sum of atom rows per molecule.

Reference: mpnn_functions/readout/graph_level_output.py:9-47.  The two Linear maps are library
GEMMs (torch); the per-molecule sum of a compact batch reuses the segmented-sum kernel with
graph_ptr as row_ptr.  Dense batches sum over dim 1 exactly as the reference does.
"""
import torch
from torch import nn

from mpnn_amd import ops
from mpnn_amd.graph import MolGraph


class GraphLevelOutput(nn.Module):
    def __init__(self, node_features, output_dim, time_steps=100, inner_prod="default", activation_fn=None,
                 attn_act=None, dropout=0):
        super().__init__()
        self.in_dim = node_features
        self.out_dim = output_dim
        self.act_fn = activation_fn() if activation_fn is not None else nn.ReLU()
        self.attn_act = attn_act() if attn_act is not None else nn.Softmax(dim=1)
        self.dropout = dropout
        self.i = nn.Sequential(nn.Linear(2 * self.in_dim, self.out_dim))
        self.j = nn.Sequential(nn.Linear(2 * self.in_dim, self.out_dim))

    def forward(self, input_set, mask=None, mprev=None, cprev=None, graph=None):
        if mask is not None:
            x = input_set * mask
            gated = torch.softmax(self.i(x), dim=-1) * self.j(x) * mask
        else:
            gated = torch.softmax(self.i(input_set).sum(dim=1), dim=-1).unsqueeze(1) * self.j(input_set)
        if isinstance(graph, MolGraph) and gated.dim() == 2:
            return ops.molecule_sum(gated, graph)
        return gated.sum(dim=1)
