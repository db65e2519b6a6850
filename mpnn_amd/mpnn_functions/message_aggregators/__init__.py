"""Aggregators of per-pair messages into per-atom rows (operator slot `message_agg_func` of the models).

All three reduce to the destination-sorted segmented sum kernel (mpnn_segsum_f32); they differ in the per-pair
weight and in whether non-bonded pairs of the padded row take part:

    AdjMsgAgg    weight = adj value, member pairs only
    WAdjMsgAgg   softmax over the padded adjacency row, every pair of the row
    AttMsgAgg    act(Linear(adj)), every pair of the row
"""
from . import adjacent_message_agg as _adj
from . import attention_message_agg as _att
from . import weighted_adjacent_message_agg as _wadj

AdjMsgAgg = _adj.AdjMsgAgg
WAdjMsgAgg = _wadj.WAdjMsgAgg
AttMsgAgg = _att.AttMsgAgg

__all__ = ["AdjMsgAgg", "WAdjMsgAgg", "AttMsgAgg"]
