"""Attention model: one AttEdgeNetwork per step, AdjMsgAgg, GRU, parameter-free masked norm.
Reference: models/att_model.py:6-59 (default readout Set2Vec, as there)."""
import torch
from torch import nn

from mpnn_amd.mpnn_functions import AdjMsgAgg, AttEdgeNetwork, GRUUpdate, Set2Vec
from ._batch import graph_of
from .mask_batch_norm import MaskBatchNorm


class BasicModel(nn.Module):
    def __init__(self, node_features, edge_features, message_features, adjacency_dim, output_dim,
                 message_func=AttEdgeNetwork, message_opts={},
                 message_agg_func=AdjMsgAgg, agg_opts={},
                 update_func=GRUUpdate, update_opts={}, message_steps=3,
                 readout_func=Set2Vec, readout_opts={}):
        super().__init__()
        message_opts.update(node_features=node_features, edge_features=edge_features,
                            message_features=message_features)
        agg_opts.update(adj_dim=adjacency_dim)
        update_opts.update(node_features=node_features, message_features=message_features)
        readout_opts.update(node_features=node_features, output_dim=output_dim)

        self.out_dim = output_dim
        self.iters = message_steps
        self.mfs = []
        for i in range(message_steps):
            mf = message_func(**message_opts)
            if hasattr(mf, "pairwise"):
                mf.pairwise = True
            self.mfs.append(mf)
            self.add_module('mf' + str(i), mf)
        self.ma = message_agg_func(**agg_opts)
        self.uf = update_func(**update_opts)
        self.of = readout_func(**readout_opts)
        self.bn = MaskBatchNorm()

    def message_passing(self, afm, bfm, adj, mask):
        graph = graph_of(afm, bfm, adj)
        node_state = afm
        for mf in self.mfs:
            if hasattr(mf, "bind_graph"):
                mf.bind_graph(graph)
            node_state = self.bn(self.uf(self.ma(mf(afm, bfm), adj), node_state, mask), mask)
        return node_state, graph

    def forward(self, afm, bfm, adj, mask):
        node_state, graph = self.message_passing(afm, bfm, adj, mask)
        readout_in = torch.cat([node_state, afm], dim=-1)
        if readout_in.dim() == 2:
            return self.of(readout_in, mask=mask, graph=graph)
        return self.of(readout_in, mask=mask)
