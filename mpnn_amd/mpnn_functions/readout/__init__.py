from .graph_level_output import GraphLevelOutput
from .set2vec import Set2Vec

__all__ = ["GraphLevelOutput", "Set2Vec"]
