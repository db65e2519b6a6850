"""Readouts (operator slot `readout_func`): per-atom rows -> one row per molecule.

    GraphLevelOutput   softmax-gated sum (two Linear maps + the segmented sum over graph_ptr)
    Set2Vec            attention read driven by an input-free LSTM cell (LSTMCellHidden)
"""
from . import graph_level_output as _glo
from . import set2vec as _s2v

GraphLevelOutput = _glo.GraphLevelOutput
Set2Vec = _s2v.Set2Vec
LSTMCellHidden = _s2v.LSTMCellHidden

__all__ = ["GraphLevelOutput", "Set2Vec", "LSTMCellHidden"]
