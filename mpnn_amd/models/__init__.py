"""Model assemblies of the reference's `models` package on the MI355X operators:

    basic_model.BasicModel, graph_model_wrapper.GraphWrapper      the north-star path
    lipo_basic_model, graph_norm_wrapper, batch_norm_graph_wrapper, mask_batch_norm, att_model

Import the modules by name, as the reference's drivers do (`from models.basic_model import BasicModel`).
"""
