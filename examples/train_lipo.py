#!/usr/bin/env python3
"""Python-3 reproduction of the reference's lipophilicity driver on the HIP path.

Assembly as in test_lipo.py:100-152 of the reference: atom features = 19 categorical columns + 3 numeric
columns that the wrapper batch-normalises (masked), nf = mf = 22, readout width 2*19, a dense head that
halves from 38 until <= 10 and then maps to 1, `BasicModel.init_weights`, seed 317, Adam(lr 1e-2,
weight_decay 1e-4) + ReduceLROnPlateau, MSE loss, batches of 16 molecules, 6 message-passing steps.
RDKit featurisation and the Lipophilicity CSV are not available here (no rdkit, no network), so the
molecules are synthetic graphs of the same shape with a synthetic regression target.

    python examples/train_lipo.py [--mols 512] [--epochs 3]
"""
import argparse
import os
import sys

import numpy as np
import torch
from torch import nn, optim

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import synth  # noqa: E402
from mpnn_amd.models.graph_norm_wrapper import GraphWrapper  # noqa: E402
from mpnn_amd.models.lipo_basic_model import BasicModel  # noqa: E402

AF, NAF, EF = 19, 3, 7


def dense_head(width):
    layers, den = [], width
    while den > 10:
        new_den = int(np.ceil(den / 2))
        layers += [nn.Linear(den, new_den), nn.ReLU()]
        den = new_den
    layers.append(nn.Linear(den, 1))
    return nn.Sequential(*layers)


def build_model(message_steps=6):
    model = nn.Sequential(
        GraphWrapper(BasicModel(AF + NAF, EF, AF + NAF, 50, 2 * AF, message_opts={}, agg_opts={}, update_opts={},
                                readout_opts={}, message_steps=message_steps), NAF),
        nn.BatchNorm1d(2 * AF),
        dense_head(2 * AF),
    )
    model.float()
    model.apply(BasicModel.init_weights)
    return model


def make_batches(num_mols, batch_size, seed, device):
    mb = synth.make_molecules(num_mols, AF + NAF, seed=seed, dist="lipo", edge_features=EF, lipo_features=True)
    rng = np.random.default_rng(seed)
    w = rng.standard_normal(AF + NAF).astype(np.float32)
    batches = []
    for b0 in range(0, num_mols - batch_size + 1, batch_size):
        sub = synth.select(mb, np.arange(b0, b0 + batch_size))
        d = synth.to_dense(sub, numeric_tail=NAF)
        # synthetic "lipophilicity": a smooth function of the molecule's mean atom features
        per_mol = np.add.reduceat(sub.atom_feat @ w, sub.atom_ptr[:-1]) / sub.n_atoms
        d["labels"] = np.tanh(per_mol).astype(np.float32)
        batches.append({k: torch.from_numpy(v).to(device) for k, v in d.items()})
    return batches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=512)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "the HIP path has no CPU fallback"
    dev = torch.device("cuda:0")
    torch.manual_seed(317)
    model = build_model(args.steps).to(dev)
    print("Model has: %d parameters" % sum(p.numel() for p in model.parameters()))
    batches = make_batches(args.mols, 16, 317, dev)
    n_val = max(1, len(batches) // 10)
    train, val = batches[n_val:], batches[:n_val]
    criterion = nn.MSELoss()
    optimizer = optim.Adam(model.parameters(), lr=1e-2, weight_decay=1e-4)
    scheduler = optim.lr_scheduler.ReduceLROnPlateau(optimizer)
    for epoch in range(args.epochs):
        model.train()
        tot = 0.0
        for batch in train:
            model.zero_grad()
            loss = criterion(model(batch), batch["labels"].float().unsqueeze(-1))
            tot += loss.item()
            loss.backward()
            optimizer.step()
        model.eval()
        with torch.no_grad():
            v = sum(criterion(model(b), b["labels"].float().unsqueeze(-1)).item() for b in val) / len(val)
        scheduler.step(v)
        print("epoch %d train loss %.5f val mse %.5f" % (epoch, tot / len(train), v))


if __name__ == "__main__":
    main()
