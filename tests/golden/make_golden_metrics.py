"""G7: golden vectors for the metric functions of the reference (ehf.compute_f1:530,
compute_MAP_MRR:714 with get_MAP:704 / get_MRR:684 / get_row_MRR:669), produced by importing the
real ehf (build container only).  `np.float` was removed from numpy; the reference still uses it
(ehf:678), so it is shimmed here exactly as SURVEY §8c notes."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for m in ("torchvision", "torchvision.datasets"):
    sys.modules.setdefault(m, types.ModuleType(m))
sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
sys.path.insert(0, "/root/reference/TensorGCN-master")
np.float = float  # noqa: the shim
import embedding_help_functions as ehf  # noqa: E402

out = {}
for case, (T, N, E, seed, dup) in enumerate([(4, 60, 500, 0, False), (3, 25, 300, 1, True), (1, 10, 40, 2, False)]):
    g = torch.Generator().manual_seed(seed)
    edges = torch.stack([torch.randint(0, T, (E,), generator=g), torch.randint(0, N, (E,), generator=g),
                         torch.randint(0, N, (E,), generator=g)])
    if not dup:  # unique (t, i, j)
        key = (edges[0] * N + edges[1]) * N + edges[2]
        _, first = np.unique(key.numpy(), return_index=True)
        edges = edges[:, torch.from_numpy(np.sort(first))]
    E2 = edges.shape[1]
    logits = torch.randn(E2, 2, generator=g) * 2
    target = (torch.rand(E2, generator=g) < 0.8).long()  # 0 = existing edge (minority in the scripts; any mix works)
    guess = logits.argmax(1)
    p, r, f1 = ehf.compute_f1(guess, target)
    MAP, MRR = ehf.compute_MAP_MRR(logits, target, edges)
    out.update({f"c{case}_edges": edges.numpy(), f"c{case}_logits": logits.numpy(), f"c{case}_target": target.numpy(),
                f"c{case}_f1": np.array([float(p), float(r), float(f1)]), f"c{case}_map": float(MAP), f"c{case}_mrr": float(MRR)})
    print(case, E2, float(p), float(r), float(f1), float(MAP), float(MRR))
out["n_cases"] = 3
np.savez_compressed(os.path.join(HERE, "g7_metrics.npz"), **out)
