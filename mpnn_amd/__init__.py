"""mpnn_amd: the message-passing hot path of hochshi/mpnn on MI355X (gfx950).

    mpnn_amd.mpnn_functions   operator API (EdgeNetwork, AdjMsgAgg, GRUUpdate, ...)
    mpnn_amd.models           basic_model / graph_model_wrapper / lipo_basic_model / att_model
    mpnn_amd.graph.MolGraph   sparse batch (CSR by destination) resident in HBM
    mpnn_amd.ops              autograd bindings of the HIP kernels (C ABI: include/mpnn_amd.h)

Importing the package needs neither a GPU nor the built library; the first kernel call loads
mpnn_amd/lib/libmpnn_amd.so and raises if it is missing (there is no CPU fallback).
"""
__version__ = "0.3.0"
