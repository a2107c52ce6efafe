#!/usr/bin/env python3
"""End-to-end example in the shape of the reference's experiment scripts
(experiment_bitcoin_our.py / experiment_reddit_our_link_prediction.py), on a synthetic dynamic
graph because no dataset ships with the reference and there is no network:

    raw edges --(device adjacency pipeline)--> Ĉ, Â = M x1 Ĉ
    gcn = EmbeddingGCN2(Â, X, edges, M, hidden_feat=[6,6,2], condensed_W=True, use_Minv=False, nonlin2="selu")
    SGD(lr .01, momentum .9) on the class-weighted cross entropy, F1 of the minority class every 100 epochs

    python examples/experiment_synthetic_our.py [--epochs 1000] [--graph]   # --graph: whole epoch as one hipGraph
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmgcn_amd.layers as ehf  # drop-in for `import embedding_help_functions as ehf`
from tmgcn_amd import WeightedCrossEntropy, metrics, synth
from tmgcn_amd.adjacency import build_adjacency
from tmgcn_amd.graphs import GraphedTrainStep

# Settings (the reference keeps these as constants at the top of each script)
S_train, N, edges_per_slice = 95, 6000, 250
no_diag, edge_life_window = 20, 10
alpha, lr, momentum = 0.9, 0.01, 0.9


def f1_minority(logits, target):
    """precision / recall / F1 with class 0 as the positive class (ehf.compute_f1, on the device)."""
    return tuple(float(x) for x in metrics.compute_f1(logits.argmax(dim=1), target))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    rng = np.random.default_rng(0)

    # raw dynamic graph: (slice, src, dst) with unit weights; labels depend weakly on node parity
    t = np.repeat(np.arange(S_train), edges_per_slice)
    i = rng.integers(0, N, t.size)
    j = rng.integers(0, N, t.size)
    labels = ((i + j) % 2 == 0) ^ (rng.random(t.size) < 0.2)
    target = torch.from_numpy(labels.astype(np.int64)).cuda()

    tic = time.perf_counter()
    M = synth.band_M(S_train, no_diag, "matlab")
    Chat, Ahat = build_adjacency(t, i, j, np.ones(t.size, np.float32), S_train, N, M=M, window=edge_life_window)
    torch.cuda.synchronize()
    print(f"adjacency pipeline on the device: nnz(C)={Chat.nnz}, nnz(A)={Ahat.nnz}, {1e3 * (time.perf_counter() - tic):.1f} ms")

    # node features = in/out degree of the raw graph (ehf.create_node_features)
    X = torch.zeros(S_train, N, 2)
    X[:, :, 0].index_put_((torch.from_numpy(t), torch.from_numpy(j)), torch.ones(t.size), accumulate=True)
    X[:, :, 1].index_put_((torch.from_numpy(t), torch.from_numpy(i)), torch.ones(t.size), accumulate=True)
    edges = torch.from_numpy(np.stack([t, i, j]))

    torch.manual_seed(0)
    gcn = ehf.EmbeddingGCN2(Ahat, X, edges, M, hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False, nonlin2="selu")
    optimizer = torch.optim.SGD(gcn.parameters(), lr=lr, momentum=momentum)
    criterion = WeightedCrossEntropy(torch.tensor([alpha, 1.0 - alpha])).cuda()  # nn.CrossEntropyLoss(weight=...) works too
    step = GraphedTrainStep(gcn, criterion, optimizer, target, keep_logits=True) if args.graph else None   # the loop reads step.output

    torch.cuda.synchronize()
    tic = time.perf_counter()
    for ep in range(args.epochs):
        if step is not None:
            loss, output = step(), step.output
        else:
            optimizer.zero_grad()
            output = gcn()
            loss = criterion(output, target)
            loss.backward()
            optimizer.step()
        if ep % 100 == 0:
            with torch.no_grad():
                p, r, f1 = f1_minority(output, target)
            print(f"ep {ep:5d}  loss {float(loss):.6f}  precision/recall/f1 {p:.4f}/{r:.4f}/{f1:.4f}")
    torch.cuda.synchronize()
    print(f"{args.epochs} epochs in {time.perf_counter() - tic:.2f} s ({'hipGraph replay' if args.graph else 'eager'})")


if __name__ == "__main__":
    main()
