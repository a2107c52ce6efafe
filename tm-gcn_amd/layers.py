"""Drop-in TM-GCN modules: same class names, constructor arguments, parameter names, parameter
draw order and ``__call__`` contract as the reference's ``embedding_help_functions``
(``import tmgcn_amd.layers as ehf``), computed by the HIP kernels on an MI355X.

    EmbeddingGCN    ehf:156-234   1-layer TM-GCN
    EmbeddingGCN2   ehf:236-357   2-layer TM-GCN (three layer-2 variants)
    EmbeddingKWGCN  ehf:425-497   baseline GCN without the M-product, 1 or 2 layers
    EmbeddingGCN_reg ehf:359-423  1-layer TM-GCN + per-node linear regression head (SEIR scripts)

Contract kept from the reference
  * ``At`` is a Python list of T sparse COO matrices (or an already built ``BatchedCSR``),
    ``X`` a dense [T,N,F0] tensor, ``edges`` an int64 [3,E] tensor of (slice, src, dst),
    ``M`` a [T,T] matrix.  Inputs may live on the CPU in fp64 as in the reference scripts;
    they are moved to the device in fp32 once.
  * ``gcn()`` uses the tensors cached at construction; ``gcn(At_list, X, edges)`` recomputes
    (the reference's ``type(At)==list`` rule, ehf:212, 316, 476).
  * Parameters are drawn with ``t.randn`` on the CPU generator in the reference's order
    (W, U / W1, W2, U / (W2), W1, U) so a seeded script starts from the same weights.
  * ``EmbeddingGCN2`` layer 2 always uses the *training* adjacency ``self.At`` (ehf:339, 343,
    348), also in validation/test calls.
  * Returns fp32 logits [E, C] on the device.  The classes exported by ``tmgcn_amd.ehf`` set
    ``host_operands`` so that a script which keeps its targets and criterion on the host runs
    unchanged (hosted.DeviceResult); ``output_device = "cpu"`` copies the result to the host instead.
The reference computes P1/P2 in fp64 and rounds to fp32 (ehf:205); here everything is fp32,
within the stated tolerance 1e-5·max|ref|.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch
import torch.nn as nn

from . import ops
from .csr import BatchedCSR

AdjLike = Union[Sequence[torch.Tensor], BatchedCSR]


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("tmgcn_amd needs a ROCm device (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _adj(At: AdjLike, N: int, device) -> BatchedCSR:
    if isinstance(At, BatchedCSR):
        if At.N != N:
            raise RuntimeError(f"adjacency has N={At.N} but X has {N} nodes")
        return At.to(device)
    return BatchedCSR.from_coo_list(list(At), N=N, device=device)


def _feat(X: torch.Tensor, device) -> torch.Tensor:
    return X.detach().to(device=device, dtype=torch.float32).contiguous()


def _is_recompute_call(At, X, edges) -> bool:
    # ehf:212 — `type(At)==list and type(X)==t.Tensor and type(edges)==t.Tensor`
    return (type(At) == list or isinstance(At, BatchedCSR)) and type(X) == torch.Tensor and type(edges) == torch.Tensor


_EdgeIndex = ops.EdgeIndex


def _edge_head(Z: torch.Tensor, idx, U: torch.Tensor) -> torch.Tensor:
    # ehf:228-232 / 351-355 / 491-495 (P4)
    return ops.edge_head(Z, idx, U)


_NONLIN = ("relu", "leaky", "selu")


class _Deliver:
    """Mixin: how ``forward`` hands its result over.
    output_device   None = leave it on the compute device; "cpu" = autograd-aware copy to the host.
    host_operands   wrap the (device) result as hosted.DeviceResult, so that the host tensors a
                    reference script combines it with (targets, class weights) follow it to the device."""
    output_device = None
    host_operands = False

    def _deliver(self, out: torch.Tensor) -> torch.Tensor:
        if self.output_device is not None:
            out = out.to(self.output_device)
        if self.host_operands:
            from .hosted import DeviceResult
            out = out.as_subclass(DeviceResult)
        return out


def _param(t: torch.Tensor, dev, dtype) -> nn.Parameter:
    """A parameter drawn on the CPU generator (reference order/values), stored on the device in
    `dtype` (fp32, or bf16 for the "bf16 weights" config)."""
    return nn.Parameter(t.to(device=dev, dtype=dtype))


def _w(p: torch.Tensor) -> torch.Tensor:
    """Parameters enter the kernels in fp32 (a bf16 parameter is widened; its gradient is
    rounded back to bf16 by autograd)."""
    return p if p.dtype == torch.float32 else p.float()


class EmbeddingGCN(_Deliver, nn.Module):
    """1-layer TM-GCN (ehf:156-234)."""

    def __init__(self, At: AdjLike, X: torch.Tensor, edges: torch.Tensor, M: torch.Tensor,
                 hidden_feat=[2, 2], condensed_W=False, use_Minv=True, device=None,
                 param_dtype=torch.float32):
        super().__init__()
        dev = torch.device(device) if device is not None else _default_device()
        self.use_Minv = use_Minv
        self.T, self.N = int(X.shape[0]), int(X.shape[1])
        self.F = [int(X.shape[-1])] + list(hidden_feat)
        self.Mop = ops.MOperator(M, dev)
        if self.Mop.T != self.T:
            raise RuntimeError(f"M is {self.Mop.T}x{self.Mop.T} but X has T={self.T}")
        if use_Minv:
            self.Minv = self.Mop.inverse()
        w_shape = (self.F[0], self.F[1]) if condensed_W else (self.T, self.F[0], self.F[1])
        self.W = _param(torch.randn(*w_shape), dev, param_dtype)                      # ehf:189/191
        self.U = _param(torch.randn(2 * self.F[1], self.F[2]), dev, param_dtype)     # ehf:192
        self.AtXt = self.compute_AtXt(_adj(At, self.N, dev), _feat(X, dev))   # ehf:195
        self._edges = _EdgeIndex(edges, self.N, dev, T=self.T)
        self.dev = dev

    def compute_AtXt(self, At: BatchedCSR, X: torch.Tensor) -> torch.Tensor:
        """ehf:203-208 — P1 then P2."""
        return ops.spmm(At, ops.m_transform(X, self.Mop))

    def forward(self, At=None, X=None, edges=None):
        if _is_recompute_call(At, X, edges):
            AtXt = self.compute_AtXt(_adj(At, self.N, self.dev), _feat(X, self.dev))
            eidx = _EdgeIndex(edges, self.N, self.dev, T=self.T)
        else:
            AtXt, eidx = self.AtXt, self._edges
        Y = ops.feature_gemm(AtXt, _w(self.W))                               # ehf:222
        if self.use_Minv:
            Y = ops.m_transform(Y, self.Minv)                                # ehf:224
        return self._deliver(_edge_head(Y, eidx, _w(self.U)))


class EmbeddingGCN_reg(_Deliver, nn.Module):
    """1-layer TM-GCN with a linear regression head per node (ehf:359-423; the SEIR experiments).
    As in the reference, ``forward`` ignores its arguments and always uses the tensors cached at
    construction (ehf:410-412), and returns [T, N]."""

    def __init__(self, At: AdjLike, X: torch.Tensor, M: torch.Tensor, hidden_feat=[2, 2], condensed_W=False,
                 use_Minv=True, device=None):
        super().__init__()
        dev = torch.device(device) if device is not None else _default_device()
        self.use_Minv = use_Minv
        self.T, self.N = int(X.shape[0]), int(X.shape[1])
        self.F = [int(X.shape[-1])] + list(hidden_feat)
        self.Mop = ops.MOperator(M, dev)
        if self.Mop.T != self.T:
            raise RuntimeError(f"M is {self.Mop.T}x{self.Mop.T} but X has T={self.T}")
        if use_Minv:
            self.Minv = self.Mop.inverse()
        w_shape = (self.F[0], self.F[1]) if condensed_W else (self.T, self.F[0], self.F[1])
        self.W = _param(torch.randn(*w_shape), dev, torch.float32)          # ehf:392/394
        self.lin1 = nn.Linear(self.F[1], 1).to(dev)                         # ehf:395 (initialised on the CPU generator)
        self.AtXt = ops.spmm(_adj(At, self.N, dev), ops.m_transform(_feat(X, dev), self.Mop))  # ehf:398
        self.dev = dev

    def forward(self, At=None, X=None):
        Y = ops.feature_gemm(self.AtXt, self.W)                               # ehf:415
        if self.use_Minv:
            Y = ops.m_transform(Y, self.Minv)                                # ehf:417
        return self._deliver(self.lin1(Y).squeeze(2))                        # ehf:421-423


class EmbeddingGCN2(_Deliver, nn.Module):
    """2-layer TM-GCN (ehf:236-357)."""

    def __init__(self, At: AdjLike, X: torch.Tensor, edges: torch.Tensor, M: torch.Tensor,
                 hidden_feat=[2, 2, 2], condensed_W=False, use_Minv=True, apply_M_twice=False,
                 apply_M_three_times=False, nonlin2="relu", device=None, param_dtype=torch.float32):
        super().__init__()
        dev = torch.device(device) if device is not None else _default_device()
        if nonlin2 not in _NONLIN:
            raise RuntimeError(f"nonlin2 must be one of {_NONLIN}")
        self.use_Minv = use_Minv
        self.apply_M_twice = apply_M_twice
        self.apply_M_three_times = apply_M_three_times
        self.nonlin2 = nonlin2
        self.T, self.N = int(X.shape[0]), int(X.shape[1])
        self.F = [int(X.shape[-1])] + list(hidden_feat)
        self.Mop = ops.MOperator(M, dev)
        if self.Mop.T != self.T:
            raise RuntimeError(f"M is {self.Mop.T}x{self.Mop.T} but X has T={self.T}")
        if use_Minv:
            self.Minv = self.Mop.inverse()
        lead = () if condensed_W else (self.T,)
        self.W1 = _param(torch.randn(*lead, self.F[0], self.F[1]), dev, param_dtype)  # ehf:278/281
        self.W2 = _param(torch.randn(*lead, self.F[1], self.F[2]), dev, param_dtype)  # ehf:279/282
        self.U = _param(torch.randn(self.F[2] * 2, self.F[3]), dev, param_dtype)      # ehf:283
        self.At = _adj(At, self.N, dev)                                            # ehf:267
        self.AtXt = self.compute_AtXt(self.At, _feat(X, dev))                      # ehf:293
        self._edges = _EdgeIndex(edges, self.N, dev, T=self.T)
        self.dev = dev

    def compute_AX(self, A: BatchedCSR, X: torch.Tensor) -> torch.Tensor:
        """ehf:301-305 — P2 only."""
        return ops.spmm(A, X)

    def compute_AtXt(self, At: BatchedCSR, X: torch.Tensor) -> torch.Tensor:
        """ehf:307-312 — P1 then P2."""
        return ops.spmm(At, ops.m_transform(X, self.Mop))

    def forward(self, At=None, X=None, edges=None):
        if _is_recompute_call(At, X, edges):
            AtXt = self.compute_AtXt(_adj(At, self.N, self.dev), _feat(X, self.dev))
            eidx = _EdgeIndex(edges, self.N, self.dev, T=self.T)
        else:
            AtXt, eidx = self.AtXt, self._edges
        W1, W2, U = _w(self.W1), _w(self.W2), _w(self.U)
        # first layer (ehf:330-335)
        if self.use_Minv:
            Y = ops.activation(ops.m_transform(ops.feature_gemm(AtXt, W1), self.Minv), self.nonlin2)
        else:
            Y = ops.feature_gemm(AtXt, W1, act=self.nonlin2)
        # second layer — always the training adjacency self.At (ehf:339, 343, 348)
        if self.use_Minv:
            Z = ops.m_transform(ops.spmm_feature_gemm(self.At, ops.m_transform(Y, self.Mop), W2), self.Minv)
        elif self.apply_M_twice:
            Z = ops.spmm_feature_gemm(self.At, ops.m_transform(Y, self.Mop), W2)
            if self.apply_M_three_times:
                Z = ops.m_transform(Z, self.Mop)                                   # ehf:346
        else:
            Z = ops.spmm_feature_gemm(self.At, Y, W2)                              # ehf:348-349
        return self._deliver(_edge_head(Z, eidx, U))


class EmbeddingKWGCN(_Deliver, nn.Module):
    """Baseline GCN without the M-product, 1 or 2 layers (ehf:425-497)."""

    def __init__(self, A: AdjLike, X: torch.Tensor, edges: torch.Tensor, hidden_feat=[2, 2],
                 nonlin2="relu", device=None, param_dtype=torch.float32):
        super().__init__()
        dev = torch.device(device) if device is not None else _default_device()
        if nonlin2 not in _NONLIN:
            raise RuntimeError(f"nonlin2 must be one of {_NONLIN}")
        self.no_layers = len(hidden_feat) - 1
        self.nonlin2 = nonlin2
        self.T, self.N = int(X.shape[0]), int(X.shape[1])
        self.F = [int(X.shape[-1])] + list(hidden_feat)
        if self.no_layers == 2:
            self.W2 = _param(torch.randn(self.F[1], self.F[2]), dev, param_dtype)     # ehf:452 (drawn first)
        self.W1 = _param(torch.randn(self.F[0], self.F[1]), dev, param_dtype)         # ehf:453
        self.U = _param(torch.randn(self.F[-2] * 2, self.F[-1]), dev, param_dtype)    # ehf:454
        self.A = _adj(A, self.N, dev)
        if self.A.T != self.T:
            raise RuntimeError(f"adjacency has {self.A.T} slices but X has T={self.T}")
        self._edges = _EdgeIndex(edges, self.N, dev, T=self.T)
        self.AX = self.compute_AX(self.A, _feat(X, dev))                           # ehf:464
        self.dev = dev

    def compute_AX(self, A: BatchedCSR, X: torch.Tensor) -> torch.Tensor:
        """ehf:469-473 — a [self.T, N, F] buffer whose first len(A) slices are Â_k·X_k and whose
        remaining slices stay zero: the baseline scripts validate on fewer slices than they train
        on (25 vs 150), and layer 2 then runs over all T slices of the training adjacency."""
        if A.T > self.T:
            raise RuntimeError(f"adjacency has {A.T} slices but the model was built for T={self.T} (ehf:470-472)")
        if X.shape[0] < A.T:
            raise RuntimeError(f"X has {X.shape[0]} slices but the adjacency has {A.T}")
        AX = ops.spmm(A, X if X.shape[0] == A.T else X[:A.T].contiguous())
        if A.T < self.T:
            AX = torch.cat((AX, AX.new_zeros(self.T - A.T, self.N, AX.shape[-1])), dim=0)
        return AX

    def forward(self, A=None, X=None, edges=None):
        if _is_recompute_call(A, X, edges):
            AX = self.compute_AX(_adj(A, self.N, self.dev), _feat(X, self.dev))
            eidx = _EdgeIndex(edges, self.N, self.dev, T=self.T)
        else:
            AX, eidx = self.AX, self._edges
        if self.no_layers == 2:
            Y = ops.feature_gemm(AX, _w(self.W1), act=self.nonlin2)                # ehf:486
            Z = ops.spmm_feature_gemm(self.A, Y, _w(self.W2))                      # ehf:487
        else:
            Z = ops.feature_gemm(AX, _w(self.W1))                                  # ehf:489
        return self._deliver(_edge_head(Z, eidx, _w(self.U)))
