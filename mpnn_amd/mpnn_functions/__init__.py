"""Operator API of the reference (its `mpnn_functions` package exports every operator class at top level,
mpnn_functions/__init__.py:1-4), backed by the HIP kernels of libmpnn_amd.so.

Sub-packages: message, message_aggregators, update, readout -- the four operator slots of
models.basic_model.BasicModel.
"""
from . import message, message_aggregators, readout, update
from .message import AttEdgeNetwork, BiLiniearEdgeNetwork, EdgeNetwork, GGNNMsgPass
from .message_aggregators import AdjMsgAgg, AttMsgAgg, WAdjMsgAgg
from .readout import GraphLevelOutput, LSTMCellHidden, Set2Vec
from .update import GRUUpdate

__all__ = (message.__all__ + message_aggregators.__all__ + update.__all__ + readout.__all__)
