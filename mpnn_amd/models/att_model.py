"""Attention model: one AttEdgeNetwork per step, AdjMsgAgg, GRU, parameter-free masked norm.
Reference: models/att_model.py:6-59 (default readout Set2Vec, as there)."""
import torch
from torch import nn

from mpnn_amd import ops
from mpnn_amd.mpnn_functions import AdjMsgAgg, AttEdgeNetwork, GRUUpdate, Set2Vec
from ._batch import graph_of
from .mask_batch_norm import MaskBatchNorm


class BasicModel(nn.Module):
    # True: where the library has the kernels (hidden 128 / 256), the parameter-free masked norm after each update is not
    # run as passes of its own -- its moments come out of the update kernel's epilogue, it is applied where the NEXT
    # update reads its state, and its backward rides on the GRU backward kernels the same way (ops.GRUNormChain;
    # SURVEY 8 row f2).  Same function of the inputs; False = the standalone norm kernels, as at every other width.
    fuse_norm = True

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

    def _norm_fusable(self, afm):
        return (self.fuse_norm and type(self.bn) is MaskBatchNorm
                and type(self.uf) is GRUUpdate and self.uf.mf == self.uf.nf == afm.shape[-1]
                and ops.gru_norm_applies(self.uf.nf, afm))

    def _message_passing_fused_norm(self, afm, bfm, adj, mask, graph):
        """models/att_model.py:57-58.  The messages depend on the atom features only (`mf(afm, bfm)`), so all of them are
        formed first; the T updates and norms then run as one chain (ops.GRUNormChain): update t reads the RAW output
        of update t-1 with the norm folded in and emits the moments of its own output, the backward kernels do the
        same for the norms' backward; only the last norm (whose output the readout reads) has passes of its own."""
        cell = self.uf.gru_cell
        msgs = []
        for mf in self.mfs:
            if hasattr(mf, "bind_graph"):
                mf.bind_graph(graph)
            msgs.append(self.ma(mf(afm, bfm), adj).reshape(-1, cell.mf))
        out = ops.gru_norm_chain(afm.reshape(-1, cell.nf), msgs, mask.reshape(-1), cell.weight_ih, cell.weight_hh,
                                 cell.bias_ih, cell.bias_hh, eps=1e-6, flags=ops.BN_EPS_INSIDE,
                                 sync=self.bn.sync_stats)
        return out.view(afm.shape)

    def message_passing(self, afm, bfm, adj, mask):
        graph = graph_of(afm, bfm, adj)
        if self._norm_fusable(afm):
            return self._message_passing_fused_norm(afm, bfm, adj, mask, graph), graph
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
