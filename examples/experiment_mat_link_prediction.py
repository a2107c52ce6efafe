#!/usr/bin/env python3
"""The reference's two-stage link-prediction workflow, end to end on the MI355X:

    stage 1 (read_data.m / read_data.py)     raw (src, dst, label, time) rows -> saved_content_*.mat
    stage 2 (experiment_*_our_link_prediction.py)   load_data -> node features -> negative edges ->
            split -> EmbeddingGCN / EmbeddingGCN2 -> SGD on the weighted CE -> MAP / MRR

Stage 2 calls the ``ehf`` surface the way a reference script does (host-side targets, class
weights and criterion; ``import tmgcn_amd.ehf as ehf`` is the one line such a script changes) but
keeps its own bookkeeping: one record per evaluation, a JSON summary at the end.  The contract of
the untouched scripts themselves is pinned against the real reference in
tests/test_gpu_experiment.py (fixture G9).  The raw rows are synthetic (a planted-community
stream; no dataset ships with the reference, no network).

    python examples/experiment_mat_link_prediction.py [--epochs 300] [--layers 2] [--nodes 1000]
"""
import argparse
import os
import json
import random
import sys
import tempfile
import time

import numpy as np
import torch as t
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmgcn_amd.ehf as ehf  # instead of: import embedding_help_functions as ehf
from tmgcn_amd import preprocess

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=300)
ap.add_argument("--layers", type=int, default=2, choices=[1, 2])
ap.add_argument("--nodes", type=int, default=1000)
ap.add_argument("--edges-per-slice", type=int, default=1500)
ap.add_argument("--eval-every", type=int, default=100)
args = ap.parse_args()

# Settings (constants at the top of every reference script)
no_layers, no_epochs = args.layers, args.epochs
S_train, S_val, S_test = 30, 5, 5
beta1, beta2, cutoff = 4, 4, S_train + S_val + S_test
alpha, lr, momentum = 0.8, 0.01, 0.9
time_delta, edge_life_window, no_diag = 3600.0, 5, 10
t.set_num_threads(8)   # the script's own host-side tensor ops are tiny; hundreds of threads only slow them down
random.seed(0)
t.manual_seed(0)
rng = np.random.default_rng(0)

# ---- stage 1: raw rows -> .mat ---------------------------------------------------------------
N, TT = args.nodes, S_train + S_val + S_test
community = rng.integers(0, 4, N)
rows = []
for k in range(TT):
    src = rng.integers(0, N, 3 * args.edges_per_slice)
    dst = rng.integers(0, N, 3 * args.edges_per_slice)
    keep = (community[src] == community[dst]) | (rng.random(src.size) < 0.1)      # mostly intra-community
    src, dst = src[keep][:args.edges_per_slice], dst[keep][:args.edges_per_slice]
    stamp = 1_600_000_000 + k * time_delta + rng.uniform(0, time_delta, src.size)
    rows.append(np.stack([src + 1, dst + 1, np.ones(src.size), stamp], axis=1))    # 1-based ids, like the csv files
raw = np.concatenate(rows)
# first and last time stamp span TT whole windows (TT = floor((max - min)/time_delta), read_data.m:110); node N appears
raw = np.concatenate([raw, [[N, 1, 1.0, 1_600_000_000], [1, N, 1.0, 1_600_000_000 + TT * time_delta + 1]]])
tic = time.perf_counter()
content = preprocess.read_data(raw, S_train, S_val, S_test, time_delta=time_delta, edge_life_window=edge_life_window,
                               no_diag=no_diag)
data_loc = tempfile.mkdtemp() + "/"
mat_f_name = "saved_content_synthetic.mat"
preprocess.save_content(data_loc + mat_f_name, content)
print("preprocessing: %d raw rows -> %s in %.2f s (nnz C = %d, nnz Ct_train = %d)"
      % (len(raw), mat_f_name, time.perf_counter() - tic, len(content["C_vals"]), len(content["Ct_train_vals"])))

# ---- stage 2: model, training, evaluation -----------------------------------------------------
A, A_labels, Ct_train_2, Ct_val_2, Ct_test_2, N, M = ehf.load_data(data_loc, mat_f_name, S_train, S_val, S_test, transformed=True)
X_train, X_val, X_test = ehf.create_node_features(A, S_train, S_val, S_test, same_block_size=True)
edges_aug, labels = ehf.augment_edges(A_labels._indices(), N, beta1, beta2, cutoff)
(edges_train, target_train, e_train, edges_val, target_val, e_val, K_val,
 edges_test, target_test, e_test, K_test) = ehf.split_data(edges_aug, labels, S_train, S_val, S_test, same_block_size=True)

model_kw = dict(condensed_W=True, use_Minv=False)
if no_layers == 2:
    gcn = ehf.EmbeddingGCN2(Ct_train_2[:-1], X_train[:-1], e_train, M[:-1, :-1], hidden_feat=[6, 6, 2], nonlin2="selu", **model_kw)
else:
    gcn = ehf.EmbeddingGCN(Ct_train_2[:-1], X_train[:-1], e_train, M[:-1, :-1], hidden_feat=[6, 2], **model_kw)
optimizer = t.optim.SGD(gcn.parameters(), lr=lr, momentum=momentum)
criterion = nn.CrossEntropyLoss(weight=t.tensor([alpha, 1.0 - alpha]))
in_window = edges_train[0] != 0                      # the first slice has no predecessor to predict it from
y_train, pairs_train = target_train[in_window], edges_train[:, in_window]


def held_out(At_list, X, e, K, target, pairs):
    """Forward on another window; the last K labelled edges are the ones that count (split_data)."""
    logits = gcn(At_list[:-1], X[:-1], e)[-K:]
    MAP, MRR = ehf.compute_MAP_MRR(logits, target[-K:], pairs[:, -K:])
    return {"MAP": float(MAP), "MRR": float(MRR), "loss": float(criterion(logits, target[-K:]))}


history, losses = [], []
tic = time.perf_counter()
for ep in range(no_epochs):
    optimizer.zero_grad()
    logits = gcn()
    loss = criterion(logits, y_train)
    loss.backward()
    optimizer.step()
    losses.append(loss.detach())
    if ep % args.eval_every == 0 or ep == no_epochs - 1:
        with t.no_grad():
            MAP, MRR = ehf.compute_MAP_MRR(logits, y_train, pairs_train)
            rec = {"epoch": ep,
                   "train": {"MAP": float(MAP), "MRR": float(MRR), "loss": float(loss)},
                   "val": held_out(Ct_val_2, X_val, e_val, K_val, target_val, edges_val),
                   "test": held_out(Ct_test_2, X_test, e_test, K_test, target_test, edges_test)}
        history.append(rec)
        print("epoch %4d   " % ep + "   ".join("%s: MAP %.4f MRR %.4f loss %.4f" % (k, v["MAP"], v["MRR"], v["loss"])
                                                for k, v in rec.items() if k != "epoch"))
t.cuda.synchronize()
elapsed = time.perf_counter() - tic
summary = {"layers": no_layers, "epochs": no_epochs, "seconds": round(elapsed, 3), "labelled_training_edges": int(in_window.sum()),
           "logits_device": str(logits.device), "first_loss": float(losses[0]), "last_loss": float(losses[-1]), "final": history[-1]}
print("summary: " + json.dumps(summary))
with open(data_loc + "link_prediction_history.json", "w") as f:
    json.dump({"summary": summary, "history": history}, f)
assert summary["last_loss"] < summary["first_loss"], "training loss did not go down"
