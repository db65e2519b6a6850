"""Batch plumbing shared by the models: dense padded tensors or a sparse-native MolGraph."""
from mpnn_amd.graph import MolGraph


def graph_of(afm, bfm, adj):
    """The MolGraph of a batch.  Sparse-native callers pass the MolGraph itself as `bfm`/`adj`;
    dense callers pass the reference's padded tensors, converted once per batch on the device
    (mpnn_csr_count / mpnn_csr_fill)."""
    if isinstance(bfm, MolGraph):
        return bfm
    if isinstance(adj, MolGraph):
        return adj
    return MolGraph.from_dense(adj, bfm if (bfm is not None and bfm.is_floating_point() and bfm.dim() == 4)
                               else None)
