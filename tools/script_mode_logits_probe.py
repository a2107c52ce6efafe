#!/usr/bin/env python3
"""How much of a script-mode epoch is the logits the fused criterion never reads?  

An untouched reference script calls `output = gcn(); loss = criterion(output, target)` (experiment_reddit_our_link_prediction.py:
76-77).  With hosted.FUSE_HEAD_LOSS the criterion takes the one-pass head + loss kernel from what `output` was formed from,
so the logits launch(es) inside gcn() — the edge head, and for the 1-layer model the AtXt·W GEMM in front of it — produce a
tensor nobody reads in an ordinary epoch.  Forming them lazily would need a snapshot of U (and W) at call time (the scripts
read `output_train` AFTER optimizer.step(), :81/:87) — one small launch instead of one or two.  hosted.LazyLogits (round 5) avoids the snapshot: the
criterion's own launch writes the logits as a by-product.  This probe times, interleaved in one process: script mode with the
placeholder (hosted.LAZY_LOGITS = True, the default), with the logits formed by gcn() (False: round 4), and the UPPER BOUND —
gcn() returns an UNINITIALISED [E, C] tensor that carries the head (no logits launch, no by-product store).
    python tools/script_mode_logits_probe.py [S1 S2 S3]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tmgcn_amd import layers, synth  # noqa: E402


def no_logits_forward(self, At=None, X=None, edges=None):
    Z, eidx, U, fold = self._embed(At, X, edges)
    out = torch.empty(eidx.E, U.shape[-1], device=Z.device, dtype=torch.float32).requires_grad_(True)
    return self._deliver(out, (Z, eidx, U, fold))


def main():
    names = sys.argv[1:] or ["S1", "S2", "S3"]
    rec = {}
    for name in names:
        g = synth.dynamic_graph(**synth.CONFIGS[name], seed=0)
        spec = bench.EPOCH_MODELS[name]
        orig = layers._Head.forward
        r = {}
        from tmgcn_amd import hosted
        for tag, fwd, lazy in (("script_lazy", orig, True), ("script_logits_formed_by_gcn", orig, False),
                               ("script_without_logits_upper_bound", no_logits_forward, False),
                               ("script_lazy_again", orig, True), ("script_logits_formed_by_gcn_again", orig, False)):
            layers._Head.forward = fwd
            hosted.LAZY_LOGITS = lazy
            try:
                _, med, best = bench.gpu_epochs(g, spec, 50, "script")
            finally:
                layers._Head.forward = orig
                hosted.LAZY_LOGITS = True
            r[tag + "_ms"] = round(med * 1e3, 4)
            r[tag + "_ms_min_pass"] = round(best * 1e3, 4)
        rec[name] = r
        print(name, json.dumps(r), flush=True)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
