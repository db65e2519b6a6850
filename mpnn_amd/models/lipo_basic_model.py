"""The model test_lipo.py trains: fused EdgeNetwork message (no aggregator call), masked batch
norm after the message and after the GRU, 6 steps by default.
Reference: models/lipo_basic_model.py:8-107 (forward :81-86, init_weights :88-107)."""
import torch
from torch import nn

from mpnn_amd import ops
from mpnn_amd.mpnn_functions import AdjMsgAgg, EdgeNetwork, GraphLevelOutput, GRUUpdate
from mpnn_amd.mpnn_functions.message.ggnn_msg_pass import GGNNMsgPass  # noqa: F401  (re-exported like the reference)
from ._batch import graph_of
from .mask_batch_norm import MaskBatchNorm1d


class BasicModel(nn.Module):
    # True: the norm after every update (`self.bn`, lipo_basic_model.py:85) is fused into the updates where that costs the
    # update nothing (ops.gru_norm_costs_nothing: the wide kernels at hidden 128 / 256, the generic kernel at widths that run
    # on it anyway -- the lipo model's 22-38 features): its moments come out of the update kernel, it is applied where the
    # next update reads its state, its backward rides on the GRU backward (ops.GRUNormChain).  The norm on
    # the messages (`self.ma_bn`) stays a kernel pair of its own: its input comes out of the message kernel, not an update.
    fuse_norm = True

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
        if self._norm_fusable(afm):
            node_state = self._updates_with_fused_norm(afm, bfm, mask)
        else:
            node_state = afm
            for i in range(self.iters):
                message = self.ma_bn(self.mf(afm, bfm, i != 0), mask)
                node_state = self.bn(self.uf(message, node_state, mask), mask)
        readout_in = torch.cat([node_state, afm], dim=-1)
        if readout_in.dim() == 2:
            return self.of(readout_in, mask=mask, graph=graph)
        return self.of(readout_in, mask=mask)

    def _norm_fusable(self, afm):
        bn = self.bn
        return (self.fuse_norm and self.iters > 0 and type(bn) is MaskBatchNorm1d and type(self.uf) is GRUUpdate
                and not bn.sync_stats and bn.affine and self.uf.mf == self.uf.nf == afm.shape[-1]
                and (bn.training or bn.track_running_stats) and ops.gru_norm_costs_nothing(self.uf.nf, afm))

    def _updates_with_fused_norm(self, afm, bfm, mask):
        """lipo_basic_model.py:84-85: the messages first (they depend on the atom features only; `ma_bn` is called once per
        step, as there, so its running estimates move as often), then the T updates and norms as one chain."""
        bn, cell = self.bn, self.uf.gru_cell
        msgs = [self.ma_bn(self.mf(afm, bfm, i != 0), mask).reshape(-1, cell.mf) for i in range(self.iters)]
        batch_stats = bn.training or not bn.track_running_stats
        given = None if batch_stats else (bn.running_mean, bn.running_var)
        out, stats = ops.gru_norm_chain(afm.reshape(-1, cell.nf), msgs, mask.reshape(-1), cell.weight_ih, cell.weight_hh,
                                        cell.bias_ih, cell.bias_hh, weight=bn.weight, bias=bn.bias, eps=bn.eps,
                                        flags=ops.BN_MASKED_MEAN, given=given, return_stats=True)
        if batch_stats and bn.track_running_stats:
            with torch.no_grad():                      # mask_batch_norm.py:30-33, once per norm call, in order
                keep = 1 - bn.momentum
                for mean, var in stats:
                    bn.running_mean = keep * bn.running_mean + bn.momentum * mean
                    bn.running_var = keep * bn.running_var + bn.momentum * var
        return out.view(afm.shape)

    @staticmethod
    def init_weights(m):
        """kaiming-uniform Linear weights, zero biases (=> edge_map(0) == 0); the reference's
        nn.GRUCell branch never fires for its hand-written cell, nor does it here."""
        if type(m) == nn.Linear:
            nn.init.kaiming_uniform_(m.weight, nonlinearity='relu')
            if m.bias is not None:
                nn.init.zeros_(m.bias)
