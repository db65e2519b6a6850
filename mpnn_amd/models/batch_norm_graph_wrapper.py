"""GraphWrapper variant that batch-normalises atom AND bond features before the model runs.
Reference: models/batch_norm_graph_wrapper.py:5-17 (MaskBatchNorm on afm under the atom mask, and on bfm * adj under the
adjacency as mask, so non-bonded pairs stay exactly zero and keep their "no edge" meaning).

Dense batches go through the masked-norm kernels as written.  A compact batch carries its bond features as K
distinct rows plus a type id per edge, so the statistics over all E edges are weighted sums over the K rows (weights =
edges per type) and the normalised table has K rows again: the graph stays discrete-typed.
"""
import torch
from torch import nn

from mpnn_amd.graph import MolGraph
from .mask_batch_norm import MaskBatchNorm


class GraphWrapper(nn.Module):
    def __init__(self, graph_model):
        super().__init__()
        self.add_module('graph_model', graph_model)
        self.add_module('norm', MaskBatchNorm())

    def forward(self, graph_batch):
        mask = graph_batch['mask']
        afm = self.norm(graph_batch['afm'], mask)
        g = graph_batch.get('graph')
        if isinstance(g, MolGraph):
            counts = torch.bincount(g.edge_type.to(torch.int64), minlength=g.num_types).to(g.type_feat.dtype)
            E = counts.sum()
            mean = counts @ g.type_feat / E
            c = g.type_feat - mean
            var = counts @ (c * c) / E
            gn = g.with_type_feat(c / (var + 1e-6).sqrt())
            return self.graph_model.forward(afm, gn, gn, mask)
        adj = graph_batch['adj']
        bfm = self.norm(graph_batch['bfm'] * adj.unsqueeze(-1), adj)
        return self.graph_model.forward(afm, bfm, adj, mask)
