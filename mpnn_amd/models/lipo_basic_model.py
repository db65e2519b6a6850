"""The model test_lipo.py trains: fused EdgeNetwork message (no aggregator call), masked batch
norm after the message and after the GRU, 6 steps by default.
Reference: models/lipo_basic_model.py:8-107 (forward :81-86, init_weights :88-107)."""
import torch
from torch import nn

from mpnn_amd.mpnn_functions import AdjMsgAgg, EdgeNetwork, GraphLevelOutput, GRUUpdate
from mpnn_amd.mpnn_functions.message.ggnn_msg_pass import GGNNMsgPass  # noqa: F401  (re-exported like the reference)
from ._batch import graph_of
from .mask_batch_norm import MaskBatchNorm1d


class BasicModel(nn.Module):
    def __init__(self, node_features, edge_features, message_features, adjacency_dim, output_dim,
                 message_func=EdgeNetwork, message_opts={},
                 message_agg_func=AdjMsgAgg, agg_opts={},
                 update_func=GRUUpdate, update_opts={}, message_steps=6,
                 readout_func=GraphLevelOutput, readout_opts={}, atom_encoder=None, bond_encoder=None):
        super().__init__()
        message_opts.update(node_features=node_features, edge_features=edge_features,
                            message_features=message_features)
        agg_opts.update(adj_dim=adjacency_dim)
        update_opts.update(node_features=node_features, message_features=message_features)
        readout_opts.update(node_features=node_features, output_dim=output_dim)

        self.out_dim = output_dim
        self.iters = message_steps
        self.bn = MaskBatchNorm1d(node_features)
        self.ma_bn = MaskBatchNorm1d(message_features)
        self.mf = message_func(**message_opts)
        self.ma = message_agg_func(**agg_opts)      # built (state_dict parity) but never called
        self.uf = update_func(**update_opts)
        self.of = readout_func(**readout_opts)

    def forward(self, afm, bfm, adj, mask):
        graph = graph_of(afm, bfm, None)            # HEAD semantics: adj is never consulted
        if hasattr(self.mf, "bind_graph"):
            self.mf.bind_graph(graph)
        node_state = afm
        for i in range(self.iters):
            message = self.ma_bn(self.mf(afm, bfm, i != 0), mask)
            node_state = self.bn(self.uf(message, node_state, mask), mask)
        readout_in = torch.cat([node_state, afm], dim=-1)
        if readout_in.dim() == 2:
            return self.of(readout_in, mask=mask, graph=graph)
        return self.of(readout_in, mask=mask)

    @staticmethod
    def init_weights(m):
        """kaiming-uniform Linear weights, zero biases (=> edge_map(0) == 0); the reference's
        nn.GRUCell branch never fires for its hand-written cell, nor does it here."""
        if type(m) == nn.Linear:
            nn.init.kaiming_uniform_(m.weight, nonlinearity='relu')
            if m.bias is not None:
                nn.init.zeros_(m.bias)
