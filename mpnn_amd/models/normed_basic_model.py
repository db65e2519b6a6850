"""BasicModel with one message function PER step and a parameter-free masked norm after every update.
Reference: models/normed_basic_model.py:6-59 -- the structure of att_model.py with EdgeNetwork and GraphLevelOutput as
defaults (each step's `mf<i>` is its own module, so the state_dict keys are `mf0.*`, `mf1.*`, ...)."""
from mpnn_amd.mpnn_functions import AdjMsgAgg, EdgeNetwork, GraphLevelOutput, GRUUpdate
from .att_model import BasicModel as _PerStepModel


class BasicModel(_PerStepModel):
    def __init__(self, node_features, edge_features, message_features, adjacency_dim, output_dim,
                 message_func=EdgeNetwork, message_opts={},
                 message_agg_func=AdjMsgAgg, agg_opts={},
                 update_func=GRUUpdate, update_opts={}, message_steps=3,
                 readout_func=GraphLevelOutput, readout_opts={}):
        super().__init__(node_features, edge_features, message_features, adjacency_dim, output_dim,
                         message_func=message_func, message_opts=message_opts,
                         message_agg_func=message_agg_func, agg_opts=agg_opts,
                         update_func=update_func, update_opts=update_opts, message_steps=message_steps,
                         readout_func=readout_func, readout_opts=readout_opts)
