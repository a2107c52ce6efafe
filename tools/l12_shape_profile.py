#!/usr/bin/env python3
"""Forward + backward of the fused layer-1+2 operator (ops.layer12, csrc/layer12.hip) on a random batched CSR of a given
shape, for a rocprofv3 kernel-stats pass:   rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/l12_shape_profile.py T N DEG
(DEG non-zeros per row on average, columns uniform; 2 -> 6 -> 6 features, SELU)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmgcn_amd import adjacency, ops  # noqa: E402

T, N, deg = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
rng = np.random.default_rng(0)
nnz = int(T * N * deg)
A = adjacency.DeviceCOO.from_edges(rng.integers(0, T, nnz), rng.integers(0, N, nnz), rng.integers(0, N, nnz),
                                   rng.uniform(0.1, 1.0, nnz).astype(np.float32), T, N).sort_reduce().to_csr()
g = torch.Generator().manual_seed(7)
H = torch.randn(T, N, 2, generator=g).cuda()
W1 = (torch.randn(2, 6, generator=g) * 0.7).cuda().requires_grad_(True)
W2 = (torch.randn(6, 6, generator=g) * 0.7).cuda().requires_grad_(True)
dZ = torch.randn(T, N, 6, generator=g).cuda()
print(f"T={T} N={N} rows={T * N} nnz={A.nnz} avg={A.avg_nnz_per_row:.2f}")
for _ in range(reps):
    W1.grad = W2.grad = None
    ops.layer12(H, W1, "selu", A, W2, None).backward(dZ)
torch.cuda.synchronize()
