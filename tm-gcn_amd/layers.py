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

    def _deliver(self, out: torch.Tensor, head=None) -> torch.Tensor:
        if self.output_device is not None:
            out = out.to(self.output_device)
        if self.host_operands:
            from .hosted import DeviceResult
            out = out.as_subclass(DeviceResult)
            if head is not None and self.output_device is None:
                # what the logits were formed from: lets `criterion(output, target)` of an untouched script take the
                # one-pass head + loss kernel (hosted._fused_head_loss) instead of differentiating through the logits
                out._tmgcn_head = head
        return out


class _Sharding:
    """Mixin: optional slice sharding of a drop-in model over a torch.distributed process group.

    ``group=None`` (default): the whole model on this process's device, as in the reference.
    ``group=<ProcessGroup>`` (e.g. ``dist.group.WORLD``): every rank is handed the same inputs (as a
    script run under torchrun would do) and keeps only its contiguous range of frontal slices —
    its block of the adjacency, its slices of the cached ``AtXt`` and of every activation, the
    labelled edges of its slices.  Parameters stay replicated (same seeded draws on every rank;
    gradients summed over the ranks).  Collectives: one all-gather of the activation in front of
    each M / M⁻¹ product that follows the first propagation (``apply_M_twice``,
    ``apply_M_three_times``, ``use_Minv`` — the as-run default layer-2 branch needs none), the
    all-gather of the per-rank logits into the caller's [E, C] order, and the all-reduce of the
    parameter gradients.  X is the constant layer-1 input and is kept whole."""

    _shard = None

    def _init_shard(self, group):
        if group is not None:
            from .dist import SliceShard
            self._shard = SliceShard(group, self.T)
            # T < G is the same verdict on every rank, so all of them raise together (testing
            # this rank's own Tl == 0 would stop only the ranks >= T and leave the others
            # waiting in their first collective)
            if self.T < self._shard.G:
                raise RuntimeError(f"T={self.T} slices cannot be sharded over {self._shard.G} ranks")

    def _own(self, seq):
        """This rank's slices of a list of adjacency slices / a BatchedCSR / a [T,…] tensor."""
        if self._shard is None:
            return seq
        if isinstance(seq, BatchedCSR):
            # a window shorter than T (KWGCN's validation call, ehf:469-473): both ends clamp, so a
            # rank whose slices all lie behind the window gets an EMPTY shard (and goes on to the
            # same collectives as the others) instead of failing alone
            return seq.slices(min(self._shard.k0, seq.T), min(self._shard.k1, seq.T))
        return seq[self._shard.k0:self._shard.k1]

    def _mt_input(self, X: torch.Tensor, op) -> torch.Tensor:
        """P1 of the constant input (kept whole on every rank): only this rank's output slices."""
        if self._shard is None:
            return ops.m_transform(X, op)
        return ops.m_transform(X, op, row_off=self._shard.k0, col_off=0, T_out=self._shard.Tl)

    def _mt(self, Y: torch.Tensor, op) -> torch.Tensor:
        """M (or M⁻¹) applied to a slice-sharded activation: the one exchange of that layer."""
        return ops.m_transform(Y, op) if self._shard is None else self._shard.m_transform(Y, op)

    def _p(self, param: torch.Tensor, per_slice: bool = False, gemm: bool = False) -> torch.Tensor:
        """A parameter as the kernels take it: fp32, gradient summed over the ranks when sharded,
        and — for one-weight-per-slice parameters — this rank's slices.  ``gemm``: the consumer is
        the standalone P3 kernel, which takes a bf16-stored weight as it is (three plane products
        instead of six, ops.HipKernels.gemm); the sharded model still widens, so that the gradient
        is all-reduced in fp32 and rounded once."""
        wide = getattr(self, "_wide", None)
        if gemm and self._shard is None and param.dtype == torch.bfloat16:
            w = param
        elif wide and id(param) in wide:
            w = wide[id(param)]
        else:
            w = _w(param)
        if self._shard is not None:
            w = self._shard.shared(w)
            if per_slice:
                w = w[self._shard.k0:self._shard.k1].contiguous()
        return w

    def _widen(self):
        """bf16-stored parameters ("bf16 weights"): their fp32 copies for this call in ONE launch, the gradients rounded
        back in one more (ops.widen_params) — instead of a cast launch per parameter each way."""
        ps = [p for p in self.parameters(recurse=False) if p.dtype == torch.bfloat16]
        self._wide = dict(zip(map(id, ps), ops.widen_params(ps))) if (ps and self._shard is None and ps[0].is_cuda) else None

    def _edge_index(self, edges, dev):
        if self._shard is None:
            return _EdgeIndex(edges, self.N, dev, T=self.T)
        return self._shard.edge_index(edges, self.N, dev)

    def _head(self, Z: torch.Tensor, eidx, U: torch.Tensor) -> torch.Tensor:
        if self._shard is None:
            return _edge_head(Z, eidx, U)
        local_idx, counts, gather_index, mine = eidx
        return self._shard.gather_rows(_edge_head(Z, local_idx, U), counts, gather_index, mine)


def _criterion_spec(criterion, C: int, device):
    """(class weights [C] fp32, ignore_index) when `criterion` is the scripts' criterion — a class-weighted (or
    plain) cross entropy with mean reduction and class-index targets — else None."""
    from .losses import WeightedCrossEntropy
    if isinstance(criterion, torch.Tensor):
        w, ign = criterion, -100
    elif isinstance(criterion, WeightedCrossEntropy):
        w, ign = criterion.weight, criterion.ignore_index
    elif isinstance(criterion, nn.CrossEntropyLoss):
        if criterion.reduction != "mean" or getattr(criterion, "label_smoothing", 0.0) != 0.0:
            return None
        w, ign = criterion.weight, criterion.ignore_index
        if w is None:
            w = torch.ones(C)
    else:
        return None
    if w.numel() != C or (0 <= ign < C):
        return None
    w = w.detach()
    if w.device.type == "cpu":                       # a host-side criterion (the scripts'): its weights cross PCIe once
        from .hosted import _device_copy
        w = _device_copy(w, torch.device(device))
    return w.to(device=device, dtype=torch.float32).contiguous(), int(ign)


class _Head:
    """Mixin: the two ways a model's embedding leaves it.  ``_embed(At, X, edges)`` (per class) returns
    (Z, edge index, U, fold) — fold = the shared weight W when Z is still AtXt and Z·W (ehf:222) may be done by
    the consumer.
      forward(...)              logits [E, C] = [Z[src], Z[dst]]·U            (ehf:228-232 / 351-355 / 491-495)
      loss(criterion, target)   criterion(forward(...), target) in ONE launch (ops.head_loss): the per-epoch
                                statements  `output = gcn(); loss = criterion(output, target)`  of the scripts
                                (experiment_reddit_our_link_prediction.py:76-77) as one call, gradients included."""

    def _embed(self, At=None, X=None, edges=None):
        # the widened copies of bf16 parameters live for this call only (the autograd graph keeps what it needs): a
        # model that kept them until the next call would free the previous step's copies in the middle of that step —
        # under hipGraph capture that is a free of non-captured memory inside the capture (it crashed capture_end)
        self._widen()
        try:
            return self._embed_impl(At, X, edges)
        finally:
            self._wide = None

    def forward(self, At=None, X=None, edges=None):
        Z, eidx, U, fold = self._embed(At, X, edges)
        head = (Z, eidx, U, fold) if (self._shard is None and self.host_operands) else None

        def form():
            Zf = ops.feature_gemm(Z, fold) if fold is not None else Z        # ehf:222
            return self._head(Zf, eidx, U)

        if head is not None and self.output_device is None and torch.is_grad_enabled() and U.requires_grad:
            # script mode, training epoch: the criterion will take the one-pass head + loss kernel from `head` and never read
            # the logits; they are formed by that launch as a by-product, or on first use by anything else (hosted.LazyLogits)
            from . import hosted
            if hosted.FUSE_HEAD_LOSS and hosted.LAZY_LOGITS and eidx.index_dtype == torch.int32:
                key = ((fold if fold is not None else Z).shape[-1], U.shape[-1], Z.shape[-1] if fold is not None else 0)
                ok = self.__dict__.setdefault("_lazy_ok", {})
                if key not in ok:
                    ok[key] = ops.head_loss_supported(*key)
                if ok[key]:
                    def form_from(U_then, fold_then):             # the logits of the parameter VALUES gcn() ran with (no graph)
                        Zf = ops.feature_gemm(Z.detach(), fold_then) if fold_then is not None else Z.detach()
                        return self._head(Zf, eidx, U_then)
                    return hosted.LazyLogits.make(head, eidx.E, U.shape[-1], Z.device, form, form_from)
        return self._deliver(form(), head)

    def loss(self, criterion, target: torch.Tensor, At=None, X=None, edges=None, want_logits: bool = False,
             unit_grad: bool = False):
        """``criterion(self(At, X, edges), target)``; with ``want_logits`` also the logits (detached: a by-product
        for the scripts' accuracy lines).  Fused whenever the criterion is the scripts' weighted mean cross entropy
        (nn.CrossEntropyLoss(weight=...), losses.WeightedCrossEntropy, or the class-weight tensor itself), the
        model is not slice-sharded and the head is narrow (even F <= 8, C <= 4); any other case runs the
        unfused statements, same value.  ``unit_grad``: see ops.head_loss (graphs.GraphedTrainStep sets it)."""
        spec = _criterion_spec(criterion, self.F[-1], self.dev) if self._shard is None else None
        if spec is None:
            out = self(At, X, edges)
            crit = criterion if callable(criterion) else nn.CrossEntropyLoss(weight=criterion.to(out.device))
            l = crit(out, target if self.host_operands else target.to(out.device))
            return (l, out.detach()) if want_logits else l
        Z, eidx, U, fold = self._embed(At, X, edges)
        res = ops.head_loss(Z, eidx, U, target, spec[0], spec[1], want_logits, fold_W=fold, unit_grad=unit_grad)
        if want_logits:
            return self._deliver(res[0]), self._deliver(res[1])
        return self._deliver(res)


def fused_train_step(model, criterion, target: torch.Tensor, optimizer):
    """``loss = criterion(model(), target); loss.backward(); optimizer.step()`` of the folded 1-layer model
    (EmbeddingGCN, condensed W, no M⁻¹: ehf:222) as ONE launch (ops.head_loss_sgd): returns the loss of the parameters
    as they were, with W, U and the optimizer's momentum buffers updated in place and ``.grad`` set — or None when the
    combination is not the one the kernel covers (then the caller runs the three statements)."""
    from .optim import FusedSGD
    if not isinstance(model, _Head) or model._shard is not None or not isinstance(optimizer, FusedSGD) or len(optimizer.param_groups) != 1:
        return None
    spec = _criterion_spec(criterion, model.F[-1], model.dev)
    W, U = getattr(model, "W", None), getattr(model, "U", None)
    group = optimizer.param_groups[0]
    if (spec is None or W is None or U is None or W.dtype != torch.float32 or U.dtype != torch.float32
            or {id(q) for q in group["params"]} != {id(W), id(U)}):
        return None
    Z, eidx, U_used, fold = model._embed()
    if fold is None or fold.data_ptr() != W.data_ptr() or U_used.data_ptr() != U.data_ptr() or not ops.head_loss_supported(W.shape[1], U.shape[-1], Z.shape[-1]):
        return None
    mom = float(group["momentum"])
    bufs, first = [], False
    for q in (W, U):
        st = optimizer.state[q]
        if mom != 0.0 and st.get("momentum_buffer") is None:
            st["momentum_buffer"] = torch.empty_like(q, memory_format=torch.contiguous_format)
            first = True
        bufs.append(st.get("momentum_buffer") if mom != 0.0 else None)
    loss, dW, dU = ops.head_loss_sgd(Z, eidx, W.data, U.data, target, spec[0], spec[1], bufs[0], bufs[1], group["lr"], mom,
                                     group["dampening"], group["weight_decay"], group["nesterov"], group["maximize"], first)
    W.grad, U.grad = dW, dU
    return loss


def _param(t: torch.Tensor, dev, dtype) -> nn.Parameter:
    """A parameter drawn on the CPU generator (reference order/values), stored on the device in
    `dtype` (fp32, or bf16 for the "bf16 weights" config)."""
    return nn.Parameter(t.to(device=dev, dtype=dtype))


def _w(p: torch.Tensor) -> torch.Tensor:
    """Parameters enter the kernels in fp32 (a bf16 parameter is widened; its gradient is
    rounded back to bf16 by autograd)."""
    return p if p.dtype == torch.float32 else p.float()


class EmbeddingGCN(_Head, _Deliver, _Sharding, nn.Module):
    """1-layer TM-GCN (ehf:156-234).  ``group``: slice-shard the model over a process group (_Sharding)."""

    def __init__(self, At: AdjLike, X: torch.Tensor, edges: torch.Tensor, M: torch.Tensor,
                 hidden_feat=[2, 2], condensed_W=False, use_Minv=True, device=None,
                 param_dtype=torch.float32, group=None):
        super().__init__()
        dev = torch.device(device) if device is not None else _default_device()
        self.use_Minv = use_Minv
        self.condensed_W = condensed_W
        self.T, self.N = int(X.shape[0]), int(X.shape[1])
        self._init_shard(group)
        self.F = [int(X.shape[-1])] + list(hidden_feat)
        self.Mop = ops.MOperator(M, dev)
        if self.Mop.T != self.T:
            raise RuntimeError(f"M is {self.Mop.T}x{self.Mop.T} but X has T={self.T}")
        if use_Minv:
            self.Minv = self.Mop.inverse()
        w_shape = (self.F[0], self.F[1]) if condensed_W else (self.T, self.F[0], self.F[1])
        self.W = _param(torch.randn(*w_shape), dev, param_dtype)                      # ehf:189/191
        self.U = _param(torch.randn(2 * self.F[1], self.F[2]), dev, param_dtype)     # ehf:192
        self.AtXt = self.compute_AtXt(_adj(self._own(At), self.N, dev), _feat(X, dev))   # ehf:195
        self._edges = self._edge_index(edges, dev)
        self.dev = dev

    def compute_AtXt(self, At: BatchedCSR, X: torch.Tensor) -> torch.Tensor:
        """ehf:203-208 — P1 then P2 (sharded: this rank's slices of both)."""
        return ops.spmm(At, self._mt_input(X, self.Mop))

    def _embed_impl(self, At=None, X=None, edges=None):
        if _is_recompute_call(At, X, edges):
            AtXt = self.compute_AtXt(_adj(self._own(At), self.N, self.dev), _feat(X, self.dev))
            eidx = self._edge_index(edges, self.dev)
        else:
            AtXt, eidx = self.AtXt, self._edges
        if not self.use_Minv and self.condensed_W and self._shard is None and self.W.dtype == torch.float32:
            return AtXt, eidx, self._p(self.U), self.W                       # Z = AtXt·W left to the consumer
        Y = ops.feature_gemm(AtXt, self._p(self.W, per_slice=not self.condensed_W, gemm=True))   # ehf:222
        if self.use_Minv:
            Y = self._mt(Y, self.Minv)                                       # ehf:224
        return Y, eidx, self._p(self.U), None


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


class EmbeddingGCN2(_Head, _Deliver, _Sharding, nn.Module):
    """2-layer TM-GCN (ehf:236-357).  ``group``: slice-shard the model over a process group (_Sharding)."""

    def __init__(self, At: AdjLike, X: torch.Tensor, edges: torch.Tensor, M: torch.Tensor,
                 hidden_feat=[2, 2, 2], condensed_W=False, use_Minv=True, apply_M_twice=False,
                 apply_M_three_times=False, nonlin2="relu", device=None, param_dtype=torch.float32, group=None):
        super().__init__()
        dev = torch.device(device) if device is not None else _default_device()
        if nonlin2 not in _NONLIN:
            raise RuntimeError(f"nonlin2 must be one of {_NONLIN}")
        self.use_Minv = use_Minv
        self.apply_M_twice = apply_M_twice
        self.apply_M_three_times = apply_M_three_times
        self.nonlin2 = nonlin2
        self.condensed_W = condensed_W
        self.T, self.N = int(X.shape[0]), int(X.shape[1])
        self._init_shard(group)
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
        self.At = _adj(self._own(At), self.N, dev)                                 # ehf:267 (sharded: this rank's slices)
        self.AtXt = self.compute_AtXt(self.At, _feat(X, dev))                      # ehf:293
        self._edges = self._edge_index(edges, dev)
        self.dev = dev

    def compute_AX(self, A: BatchedCSR, X: torch.Tensor) -> torch.Tensor:
        """ehf:301-305 — P2 only."""
        return ops.spmm(A, X)

    def compute_AtXt(self, At: BatchedCSR, X: torch.Tensor) -> torch.Tensor:
        """ehf:307-312 — P1 then P2 (sharded: this rank's slices of both; X is the whole constant input)."""
        return ops.spmm(At, self._mt_input(X, self.Mop))

    def _embed_impl(self, At=None, X=None, edges=None):
        if _is_recompute_call(At, X, edges):
            AtXt = self.compute_AtXt(_adj(self._own(At), self.N, self.dev), _feat(X, self.dev))
            eidx = self._edge_index(edges, self.dev)
        else:
            AtXt, eidx = self.AtXt, self._edges
        ps = not self.condensed_W
        W1, W2, U = self._p(self.W1, ps, gemm=True), self._p(self.W2, ps), self._p(self.U)
        if not self.use_Minv and not self.apply_M_twice and not ps and self._shard is None:
            # the as-run default branch (ehf:330-335 + 348-349): both layers in one launch each way, the layer-1
            # activations never stored (ops.layer12; the unfused pair where the widths do not allow it)
            return ops.layer12(AtXt, self._p(self.W1), self.nonlin2, self.At, W2), eidx, U, None
        # first layer (ehf:330-335)
        if self.use_Minv:
            Y = ops.activation(self._mt(ops.feature_gemm(AtXt, W1), self.Minv), self.nonlin2)
        else:
            Y = ops.feature_gemm(AtXt, W1, act=self.nonlin2)
        # second layer — always the training adjacency self.At (ehf:339, 343, 348); sharded, the
        # M / M⁻¹ products below are the only steps that exchange activations between ranks
        if self.use_Minv:
            Z = self._mt(ops.spmm_feature_gemm(self.At, self._mt(Y, self.Mop), W2), self.Minv)
        elif self.apply_M_twice:
            Z = ops.spmm_feature_gemm(self.At, self._mt(Y, self.Mop), W2)
            if self.apply_M_three_times:
                Z = self._mt(Z, self.Mop)                                          # ehf:346
        else:
            Z = ops.spmm_feature_gemm(self.At, Y, W2)                              # ehf:348-349
        return Z, eidx, U, None


class EmbeddingKWGCN(_Head, _Deliver, _Sharding, nn.Module):
    """Baseline GCN without the M-product, 1 or 2 layers (ehf:425-497).  ``group``: slice-shard the
    model over a process group (_Sharding) — no M, so no activation exchange in any configuration."""

    def __init__(self, A: AdjLike, X: torch.Tensor, edges: torch.Tensor, hidden_feat=[2, 2],
                 nonlin2="relu", device=None, param_dtype=torch.float32, group=None):
        super().__init__()
        dev = torch.device(device) if device is not None else _default_device()
        if nonlin2 not in _NONLIN:
            raise RuntimeError(f"nonlin2 must be one of {_NONLIN}")
        self.no_layers = len(hidden_feat) - 1
        self.nonlin2 = nonlin2
        self.T, self.N = int(X.shape[0]), int(X.shape[1])
        self._init_shard(group)
        self.F = [int(X.shape[-1])] + list(hidden_feat)
        if self.no_layers == 2:
            self.W2 = _param(torch.randn(self.F[1], self.F[2]), dev, param_dtype)     # ehf:452 (drawn first)
        self.W1 = _param(torch.randn(self.F[0], self.F[1]), dev, param_dtype)         # ehf:453
        self.U = _param(torch.randn(self.F[-2] * 2, self.F[-1]), dev, param_dtype)    # ehf:454
        n_slices = A.T if isinstance(A, BatchedCSR) else len(A)
        if n_slices != self.T:
            raise RuntimeError(f"adjacency has {n_slices} slices but X has T={self.T}")
        self.A = _adj(self._own(A), self.N, dev)
        self._edges = self._edge_index(edges, dev)
        self.AX = self.compute_AX(self.A, self._own(_feat(X, dev)))                # ehf:464
        self.dev = dev

    def compute_AX(self, A: BatchedCSR, X: torch.Tensor) -> torch.Tensor:
        """ehf:469-473 — a [self.T, N, F] buffer whose first len(A) slices are Â_k·X_k and whose
        remaining slices stay zero: the baseline scripts validate on fewer slices than they train
        on (25 vs 150), and layer 2 then runs over all T slices of the training adjacency."""
        T_mine = self.T if self._shard is None else self._shard.Tl      # A and X are this rank's slices already
        if A.T > T_mine:
            raise RuntimeError(f"adjacency has {A.T} slices but the model was built for T={self.T} (ehf:470-472)")
        if X.shape[0] < A.T:
            raise RuntimeError(f"X has {X.shape[0]} slices but the adjacency has {A.T}")
        if A.T == 0:
            return X.new_zeros(T_mine, self.N, X.shape[-1])
        AX = ops.spmm(A, X if X.shape[0] == A.T else X[:A.T].contiguous())
        if A.T < T_mine:
            AX = torch.cat((AX, AX.new_zeros(T_mine - A.T, self.N, AX.shape[-1])), dim=0)
        return AX

    def _embed_impl(self, A=None, X=None, edges=None):
        if _is_recompute_call(A, X, edges):
            n_call = A.T if isinstance(A, BatchedCSR) else len(A)
            if n_call > self.T:
                raise RuntimeError(f"adjacency has {n_call} slices but the model was built for T={self.T} (ehf:470-472)")
            A_mine = self._own(A)
            A_csr = _adj(A_mine, self.N, self.dev) if (A_mine.T if isinstance(A_mine, BatchedCSR) else len(A_mine)) else \
                BatchedCSR(torch.zeros(1, dtype=torch.int64, device=self.dev), torch.zeros(0, dtype=torch.int32, device=self.dev),
                           torch.zeros(0, dtype=torch.float32, device=self.dev), 0, self.N)
            AX = self.compute_AX(A_csr, self._own(_feat(X, self.dev)))
            eidx = self._edge_index(edges, self.dev)
        else:
            AX, eidx = self.AX, self._edges
        if self.no_layers == 2 and self._shard is None:
            Z = ops.layer12(AX, self._p(self.W1), self.nonlin2, self.A, self._p(self.W2))   # ehf:486-487 in one launch each way
        elif self.no_layers == 2:
            Y = ops.feature_gemm(AX, self._p(self.W1, gemm=True), act=self.nonlin2)  # ehf:486
            Z = ops.spmm_feature_gemm(self.A, Y, self._p(self.W2))                 # ehf:487
        else:
            Z = ops.feature_gemm(AX, self._p(self.W1, gemm=True))                  # ehf:489
        return Z, eidx, self._p(self.U), None
