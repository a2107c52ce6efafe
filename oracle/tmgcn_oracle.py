"""CPU ORACLE for the TM-GCN layer — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
It restates, on PyTorch-CPU, the algorithm of the reference's hot path
(/root/reference/TensorGCN-master/embedding_help_functions.py, "ehf") the way the reference
executes it: a Python list of per-slice COO matrices in fp64, one dense fp64 matmul for the
M-transform, one ``torch.sparse.mm`` per slice written into an fp32 buffer, fp32 matmuls with the
weights, autograd for the backward.  It is the checker the HIP kernels are compared with and
the CPU baseline bench.py times; it is never the thing shipped.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which were produced by importing the real ehf in the build container
(tests/golden/make_golden.py).  The reference has no tests or golden vectors of its own
(SURVEY.md §4).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

SELU = torch.nn.SELU()
ACTS = {
    "relu": torch.nn.ReLU(),                       # ehf:285
    "leaky": torch.nn.LeakyReLU(negative_slope=0.01),  # ehf:287
    "selu": SELU,                                  # ehf:289
}


# ----------------------------------------------------------------------------- primitives
def m_transform(M: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    """P1, ehf:204 / 308 / 346 / 404:  Y[k] = Σ_j M[k,j] X[j]  as one [T,T]·[T,N·F] matmul.
    dtype follows torch promotion of the operands exactly as in the reference (fp64 there)."""
    T = X.shape[0]
    return torch.matmul(M, X.reshape(T, -1)).reshape(X.size())


# The reference rounds P1/P2 results into fp32 buffers (`t.zeros(...)`, ehf:205, 302, 309, 470).
# Tests that need an fp64 "truth" of the same math (to judge fp32 reduction noise at full size)
# switch this to torch.float64; the reference-faithful value is float32.
BUFFER_DTYPE = torch.float32


def slice_spmm(A: Sequence[torch.Tensor], X: torch.Tensor) -> torch.Tensor:
    """P2, ehf:206-207 / 303-304 / 310-311 / 471-472: per-slice sparse·dense into an fp32 buffer."""
    T, N = len(A), X.shape[1]
    out = torch.zeros(T, N, X.shape[-1], dtype=BUFFER_DTYPE)  # fp32, as `t.zeros(...)` in the reference
    for k in range(T):
        out[k] = torch.sparse.mm(A[k], X[k])
    return out


def compute_AtXt(M, At, X):
    """ehf:203-208."""
    return slice_spmm(At, m_transform(M, X))


def flat_edge_index(edges: torch.Tensor, N: int):
    """ehf:196-198: row index t*N+node of the flattened [T*N, F] embedding matrix."""
    e = edges.long()
    return e[0] * N + e[1], e[0] * N + e[2]


def edge_head(Z: torch.Tensor, src, dst, U: torch.Tensor) -> torch.Tensor:
    """P4, ehf:228-232 / 351-355 / 491-495."""
    Zf = Z.reshape(-1, Z.shape[-1])
    return torch.matmul(torch.cat((Zf[src], Zf[dst]), dim=1).to(U.dtype), U)  # `.float()` in the reference (U is fp32)


def draw_params(kind: str, T: int, F: List[int], condensed_W: bool = True):
    """Parameter draws in the reference's order (CPU generator): ehf:189-192, 278-283, 451-454."""
    if kind == "gcn":
        W = torch.randn(F[0], F[1]) if condensed_W else torch.randn(T, F[0], F[1])
        U = torch.randn(2 * F[1], F[2])
        return dict(W=W, U=U)
    if kind == "gcn2":
        lead = () if condensed_W else (T,)
        W1 = torch.randn(*lead, F[0], F[1])
        W2 = torch.randn(*lead, F[1], F[2])
        U = torch.randn(F[2] * 2, F[3])
        return dict(W1=W1, W2=W2, U=U)
    if kind == "kw":
        p = {}
        if len(F) == 4:
            p["W2"] = torch.randn(F[1], F[2])
        p["W1"] = torch.randn(F[0], F[1])
        p["U"] = torch.randn(F[-2] * 2, F[-1])
        return p
    raise ValueError(kind)


# ----------------------------------------------------------------------------- model forwards
def gcn_forward(AtXt, W, U, src, dst, Minv: Optional[torch.Tensor] = None):
    """EmbeddingGCN.forward, ehf:221-234 (Minv given <=> use_Minv)."""
    Y = torch.matmul(AtXt, W)
    if Minv is not None:
        Y = m_transform(Minv, Y)
    return edge_head(Y, src, dst, U)


def gcn_reg_forward(AtXt, W, lin_weight, lin_bias, Minv: Optional[torch.Tensor] = None):
    """EmbeddingGCN_reg.forward, ehf:410-423: per-node linear head on the 1-layer embedding -> [T, N]."""
    Y = torch.matmul(AtXt, W)
    if Minv is not None:
        Y = m_transform(Minv, Y)
    return torch.nn.functional.linear(Y, lin_weight, lin_bias).squeeze(2)


def gcn2_forward(AtXt, At_train, M, W1, W2, U, src, dst, nonlin="relu", use_Minv=False,
                 apply_M_twice=False, apply_M_three_times=False, Minv=None):
    """EmbeddingGCN2.forward, ehf:325-357.  `At_train` is self.At: layer 2 always uses the
    training adjacency (ehf:339, 343, 348)."""
    act = ACTS[nonlin]
    Y1 = torch.matmul(AtXt, W1)
    if use_Minv:
        Y1 = m_transform(Minv, Y1)
    Y = act(Y1).double()                                   # ehf:335
    if use_Minv:
        Z = m_transform(Minv, torch.matmul(compute_AtXt(M, At_train, Y), W2))
    elif apply_M_twice:
        Z = torch.matmul(compute_AtXt(M, At_train, Y), W2)
        if apply_M_three_times:
            Z = m_transform(M, Z.double())                 # ehf:346
    else:
        Z = torch.matmul(slice_spmm(At_train, Y), W2)      # ehf:348-349
    return edge_head(Z, src, dst, U)


def pad_slices(AX: torch.Tensor, T: int) -> torch.Tensor:
    """ehf:470: `AX = t.zeros(self.T, self.N, F)` — a call with fewer slices than the model's T
    leaves the remaining slices zero."""
    if AX.shape[0] >= T:
        return AX
    return torch.cat((AX, AX.new_zeros((T - AX.shape[0],) + tuple(AX.shape[1:]))), dim=0)


def kwgcn_forward(AX, A_train, W1, U, src, dst, W2=None, nonlin="relu"):
    """EmbeddingKWGCN.forward, ehf:485-497.  AX may hold fewer slices than A_train (a validation
    window shorter than the training one): it is zero-padded to len(A_train) as ehf:470 does."""
    AX = pad_slices(AX, len(A_train))
    if W2 is not None:
        Y = ACTS[nonlin](torch.matmul(AX, W1)).double()
        Z = torch.matmul(slice_spmm(A_train, Y), W2)
    else:
        Z = torch.matmul(AX, W1)
    return edge_head(Z, src, dst, U)


# ----------------------------------------------------------------------------- the timed layer
def layer_fwd_bwd(M, At, X, W, dY):
    """One TM-GCN layer forward + backward the reference's way (the BASELINE metric's unit of
    work): Y = (Â ⋆ (M×₁X)) W, then autograd for dX and dW.  Returns (Y, dX, dW)."""
    X = X.detach().clone().requires_grad_(True)
    W = W.detach().clone().requires_grad_(True)
    Y = torch.matmul(compute_AtXt(M, At, X), W)
    Y.backward(dY)
    return Y.detach(), X.grad, W.grad


# ----------------------------------------------------------------------------- dense identities
def dense_layer(M, A_dense, X, W):
    """Reference-free identity used by the tests: einsum form of the same layer in fp64."""
    Xt = torch.einsum("kj,jnf->knf", M.double(), X.double())
    AX = torch.einsum("knm,kmf->knf", A_dense.double(), Xt)
    return AX @ W.double() if W.dim() == 2 else torch.einsum("knf,kfg->kng", AX, W.double())
