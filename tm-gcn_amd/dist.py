"""Frontal-slice sharding of the TM-GCN layer across the GPUs of one node (RCCL over xGMI).

Between P1 and the output of P3 the T frontal slices are independent (ehf:206-207, 222 loop
over k with no cross-k term), so rank r owns the contiguous slice range [k0_r, k1_r): its block
of the batched CSR, its rows of every [T,N,F] activation.  Only the M-transform mixes slices,
so the layer needs exactly ONE exchange, placed in front of (or behind) P1, and its adjoint in
the backward pass.  Two exchange modes (DESIGN.md §6 has the byte counts):

  "a2a"        the layer input arrives NODE-sharded ([T, N/G, F]: all slices of this rank's
               nodes).  P1 runs locally on full tube fibres (any M, dense or banded) and writes
               its output in the all-to-all send layout; one all-to-all per local slice
               re-partitions it to SLICE-sharded [T/G, N, F]; P2 and P3 are local.  Each rank
               moves (G-1)/G of ITS OWN shard, spread over all 7 xGMI links.  The exchange is
               pipelined slice by slice beside the fused kernel, whose one-slice launches
               alternate between two CU-MASKED streams that leave 32 CUs to RCCL's kernels.
  "allgather"  the layer input arrives SLICE-sharded ([T/G, N, F]); an all-gather replicates
               it, every rank transforms only its own output slices (row window of M), P2 and
               P3 are local.  Backward is a reduce-scatter.  This is the north-star's literal
               pattern.  Taken literally it materialises the whole [T,N,F] tensor on every GPU
               (131 GB at S4 / G = 8, and again as the reduce-scatter input), so the gather is
               NODE-CHUNKED and fused with its consumer: chunk c (all T slices of Nc nodes)
               is gathered on a side stream into one of two [T, Nc, F] buffers while P1
               transforms chunk c-1 straight into columns of the resident [T/G, N, F] result
               (tmgcn_mtransform_ld_f32) — the replicated tensor never exists, the bytes on the
               links are the same, and per output element the arithmetic is that of the
               unchunked form: bit-equal whenever both take the same M-transform kernel — always
               for a banded M, and for a dense one when F is a multiple of 4 (chunk boundaries
               then keep the 16-byte alignment the bf16-split kernel asks for) —, fp32-equal
               otherwise (`gather_chunk_nodes=0` keeps the unchunked form).

condensed_W (one shared weight) adds an all-reduce of dW — F0·F1 floats.
The collectives used (all_to_all_single, all_gather_into_tensor, reduce_scatter_tensor,
all_reduce) exist in both the RCCL ("nccl") and gloo backends, so the sharding logic is
covered by world_size-2 CPU tests with the kernels substituted by the oracle.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import ops
from .csr import BatchedCSR


def even_bounds(n: int, parts: int) -> List[Tuple[int, int]]:
    """Contiguous ranges of n items over `parts` ranks, sizes differing by at most one."""
    base, rem = divmod(n, parts)
    out, lo = [], 0
    for r in range(parts):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


GATHER_CHUNK_BYTES = 8 << 30   # target size of ONE [T, Nc, F] chunk buffer of the chunked all-gather


def chunk_nodes_for(T: int, N: int, F: int, elem: int = 4, target: int = GATHER_CHUNK_BYTES) -> int:
    """Nodes per chunk so that a [T, Nc, F] buffer is about `target` bytes; chunks of equal size
    (the last one may be shorter), at least 1 node, at most N."""
    per_node = max(1, T * F * elem)
    nc = max(1, min(N, target // per_node))
    n_chunks = -(-N // nc)
    return -(-N // n_chunks)


def memory_plan(exchange: str, T: int, G: int, N: int, F: int, F1: int, nnz_rank: int,
                gather_chunk_nodes: Optional[int] = None, elem: int = 4) -> dict:
    """Per-rank HBM bytes of one training step (forward + backward, X and W require grad) of
    ShardedTMGCNLayer, by component — computed from shapes alone, before anything is allocated.
    bench.py prints it for both exchange modes and refuses to start a mode that cannot fit.
      resident   adjacency (CSR + transposed CSR: 2 x (8 B/nnz + 8 B/row)), X, dY, W
      step       tensors alive at the step's peak: Xt, Y, AX (saved for dW), dXt, dX, dW scratch
      exchange   what the mode adds: a2a = send + receive-side layouts of one pass;
                 allgather = two [T, Nc, F] chunk buffers (chunked) or [T,N,F] twice (unchunked)"""
    if exchange not in ("a2a", "allgather", "none"):
        raise RuntimeError(f"unknown exchange {exchange!r}")
    Tl = T // G
    slab_in, slab_out = Tl * N * F * elem, Tl * N * F1 * elem
    rows = Tl * N
    plan = {
        "adjacency": 2 * (nnz_rank * 8 + (rows + 1) * 8),
        "X": slab_in, "dY": slab_out, "W": F * F1 * elem,
        "Xt": slab_in, "Y": slab_out, "AX": slab_in,
        "dXt": slab_in, "dX": slab_in,
        "dW_scratch": 64 << 20,
    }
    if exchange == "a2a" and G > 1:
        plan["exchange"] = 2 * slab_in          # P1 output in the send layout (+ its adjoint in backward)
        plan["exchange_note"] = "send-layout copy of the node-sharded P1 output, forward and backward"
    elif exchange == "allgather" and G > 1:
        if gather_chunk_nodes == 0:
            plan["exchange"] = 2 * T * N * F * elem
            plan["exchange_note"] = "unchunked: [T,N,F] gathered (forward) and again as the reduce-scatter input (backward)"
        else:
            nc = chunk_nodes_for(T, N, F, elem) if gather_chunk_nodes is None else max(1, min(N, gather_chunk_nodes))
            plan["exchange"] = 2 * T * nc * F * elem
            plan["exchange_note"] = f"two alternating [T, {nc}, F] chunk buffers ({-(-N // nc)} chunks)"
    else:
        plan["exchange"] = 0
        plan["exchange_note"] = "no exchange"
    plan["total"] = sum(v for k, v in plan.items() if isinstance(v, int))
    plan["total_gb"] = round(plan["total"] / 1e9, 1)
    return plan


def _loaded_hip_runtime() -> str:
    """Path of the HIP runtime THIS process already uses (torch ships its own copy: a stream must be
    created by the runtime that will launch on it, not by another libamdhip64 found on the search path)."""
    try:
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                return line.split()[-1]
    except OSError:
        pass
    return "libamdhip64.so"


def cu_mask_words(n_cu: int, cus_free: int) -> List[int]:
    """The CU mask of hipExtStreamCreateWithCUMask as 32-bit words: bit b set = CU b may be used.
    All n_cu CUs but the last `cus_free` (at least one CU always stays enabled)."""
    cus_free = max(0, min(int(cus_free), n_cu - 1))
    words = (n_cu + 31) // 32
    mask = [0xFFFFFFFF] * words
    if n_cu % 32:
        mask[-1] = (1 << (n_cu % 32)) - 1
    for b in range(n_cu - cus_free, n_cu):
        mask[b // 32] &= ~(1 << (b % 32)) & 0xFFFFFFFF
    return mask


_MASKED_STREAMS = {}     # (device index, cus_free, lane) -> torch.cuda.ExternalStream; created once per process


def cu_masked_stream(device, cus_free: int, lane: int = 0):
    """A HIP stream whose kernels may run on all CUs of `device` but the last `cus_free`
    (hipExtStreamCreateWithCUMask), as a torch stream.  The pipelined sharded layer launches its
    persistent kernels on such streams so that RCCL's kernels — 256-thread workgroups of 132 VGPRs
    and 20 KB of LDS each (rocprofv3, RCCL 2.26): one does NOT fit beside three 128-VGPR blocks of
    the fused kernel on a CU, so leaving a block slot per CU free gives them nothing — always find
    whole CUs to become resident on.  The gather is bound by requests in flight, not by CUs: giving
    up 16 of 256 CUs costs it about 1 %.
    The raw HIP streams are never destroyed, so they are created once per (device, cus_free, lane) and
    shared by every layer of the process.  Launch only COUNTER-scheduled persistent kernels on them (the fused
    SpMM+GEMM draws its tiles from a device counter): a kernel with a static blockIdx-strided schedule sizes its
    grid for all CUs and would run the blocks of the masked CUs as a serial tail."""
    import ctypes as C
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(cus_free), int(lane))
    if key in _MASKED_STREAMS:
        return _MASKED_STREAMS[key]
    mask = cu_mask_words(torch.cuda.get_device_properties(device).multi_processor_count, cus_free)
    words = len(mask)
    hip = C.CDLL(_loaded_hip_runtime())
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    handle = C.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(handle), words, (C.c_uint32 * words)(*mask))
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed (status {rc})")
    _MASKED_STREAMS[key] = torch.cuda.ExternalStream(handle.value, device=device)
    return _MASKED_STREAMS[key]


def _world(group):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


# ---------------------------------------------------------------------------------------
# exchanges (autograd-aware)
# ---------------------------------------------------------------------------------------
def _a2a_node_to_slice(send: torch.Tensor, Tl: int, N: int, group) -> torch.Tensor:
    """send: [Tl][G][Nl][F] (block kk = row kk of every rank's slice range) -> [Tl][N][F]."""
    G = dist.get_world_size(group)
    Nl, F = send.shape[-2], send.shape[-1]
    out = torch.empty(Tl, N, F, dtype=send.dtype, device=send.device)
    for kk in range(Tl):
        dist.all_to_all_single(out[kk].view(G, Nl, F), send[kk], group=group)
    return out


def _a2a_slice_to_node(x: torch.Tensor, G: int, group) -> torch.Tensor:
    """x: [Tl][N][F] slice-sharded -> [Tl][G][Nl][F] in the group-interleaved layout."""
    Tl, N, F = x.shape
    Nl = N // G
    out = torch.empty(Tl, G, Nl, F, dtype=x.dtype, device=x.device)
    for kk in range(Tl):
        dist.all_to_all_single(out[kk], x[kk].view(G, Nl, F), group=group)
    return out


class _NodeToSlice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, send, Tl, N, group):
        ctx.group, ctx.G = group, dist.get_world_size(group)
        return _a2a_node_to_slice(send, Tl, N, group)

    @staticmethod
    def backward(ctx, d):
        return _a2a_slice_to_node(d.contiguous(), ctx.G, ctx.group), None, None, None


class _SliceToNode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group, ctx.G = group, dist.get_world_size(group)
        ctx.N = x.shape[1]
        return _a2a_slice_to_node(x.contiguous(), ctx.G, group)

    @staticmethod
    def backward(ctx, d):
        return _a2a_node_to_slice(d.contiguous(), d.shape[0], ctx.N, ctx.group), None


class _AllGatherSlices(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        G = dist.get_world_size(group)
        out = torch.empty((G * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, d):
        G = dist.get_world_size(ctx.group)
        out = torch.empty((d.shape[0] // G,) + tuple(d.shape[1:]), dtype=d.dtype, device=d.device)
        dist.reduce_scatter_tensor(out, d.contiguous(), op=dist.ReduceOp.SUM, group=ctx.group)
        return out, None


class _ChunkedGatherTransform(torch.autograd.Function):
    """"allgather" mode, node-chunked: Xt[kk] = Σ_j M[k0+kk][j] · X_full[j] without ever holding
    X_full.  Forward, per chunk of Nc nodes: one all-gather per local slice into a [T/G][G][Nc][F]
    buffer (the group-interleaved row order tmgcn_mtransform takes as `x_group_rows`: no staging
    copy on either side), then the row window of M applied to the chunk and written into columns
    [c0, c1) of the resident result.  Backward, per chunk: Mᵀ applied to the same columns of the
    upstream gradient into a [T/G][G][Nc][F] buffer, one reduce-scatter per local slice straight
    into dX[kk, c0:c1].  Two buffers alternate, collectives run on the layer's side stream:
    chunk c+1 is on the links while chunk c is in the M-transform.  (ehf:204, 308: the statement
    being sharded is t.matmul(self.M, X.reshape(self.T,-1)).)"""

    @staticmethod
    def forward(ctx, X, layer):
        G, Tl, T, N = layer.G, layer.Tl, layer.T, layer.N
        F = X.shape[2]
        K = ops.kernels
        X = X.contiguous()
        Xt = torch.empty(Tl, N, F, dtype=X.dtype, device=X.device)
        chunks = layer.gather_chunks()
        bufs = layer.gather_buffers(F, X.dtype, X.device)
        use_streams = X.is_cuda
        if use_streams:
            main = torch.cuda.current_stream(X.device)
            comm = layer.comm_stream()
            comm.wait_stream(main)              # X is complete; the buffers' previous users are done
            done = []
        for ci, (c0, c1) in enumerate(chunks):
            nc = c1 - c0
            buf = bufs[ci % 2][:Tl * G * nc * F].view(Tl, G, nc, F)
            if use_streams:
                with torch.cuda.stream(comm):
                    if ci >= 2:
                        comm.wait_event(done[ci - 2])   # the transform that last read this buffer
                    for kk in range(Tl):
                        dist.all_gather_into_tensor(buf[kk].view(G * nc, F), X[kk, c0:c1], group=layer.group)
                    ev = torch.cuda.Event()
                    ev.record(comm)
                main.wait_event(ev)
            else:
                for kk in range(Tl):
                    dist.all_gather_into_tensor(buf[kk].view(G * nc, F), X[kk, c0:c1], group=layer.group)
            # logical input slice j = r*Tl + kk sits at storage row kk*G + r: x_group_rows = Tl
            K.mtransform_out(layer.Mop, buf.view(T, nc, F), Xt[:, c0:c1], row_off=layer.k0, col_off=0, x_group_rows=Tl)
            if use_streams:
                ev = torch.cuda.Event()
                ev.record(main)
                done.append(ev)
        ctx.layer, ctx.F = layer, F
        return Xt

    @staticmethod
    def backward(ctx, dXt):
        layer, F = ctx.layer, ctx.F
        G, Tl, T, N = layer.G, layer.Tl, layer.T, layer.N
        K = ops.kernels
        dXt = dXt.contiguous()
        dX = torch.empty(Tl, N, F, dtype=dXt.dtype, device=dXt.device)
        chunks = layer.gather_chunks()
        bufs = layer.gather_buffers(F, dXt.dtype, dXt.device)
        use_streams = dXt.is_cuda
        if use_streams:
            main = torch.cuda.current_stream(dXt.device)
            comm = layer.comm_stream()
            main.wait_stream(comm)              # nothing of an earlier pass still uses the buffers
            comm.wait_stream(main)              # dX exists before the first reduce-scatter writes it
            sent = []
        for ci, (c0, c1) in enumerate(chunks):
            nc = c1 - c0
            buf = bufs[ci % 2][:Tl * G * nc * F].view(Tl, G, nc, F)
            if use_streams and ci >= 2:
                main.wait_event(sent[ci - 2])           # the reduce-scatter that last read this buffer
            # adjoint of the row window: all T rows of Mᵀ[:, k0:k0+Tl] · dXt, in the send layout
            K.mtransform_out(layer.Mop, dXt[:, c0:c1], buf.view(T, nc, F), transpose=True, row_off=0, col_off=layer.k0,
                             y_group_rows=Tl)
            if use_streams:
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    for kk in range(Tl):
                        dist.reduce_scatter_tensor(dX[kk, c0:c1], buf[kk].view(G * nc, F), op=dist.ReduceOp.SUM, group=layer.group)
                    ev2 = torch.cuda.Event()
                    ev2.record(comm)
                    sent.append(ev2)
            else:
                for kk in range(Tl):
                    dist.reduce_scatter_tensor(dX[kk, c0:c1], buf[kk].view(G * nc, F), op=dist.ReduceOp.SUM, group=layer.group)
        if use_streams:
            main.wait_stream(comm)
        return dX, None


class _SharedWeight(torch.autograd.Function):
    """Identity on a replicated weight; sums its gradient over the ranks."""

    @staticmethod
    def forward(ctx, w, group):
        ctx.group = group
        return w.view_as(w)

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous().clone()
        dist.all_reduce(d, op=dist.ReduceOp.SUM, group=ctx.group)
        return d, None


class _PipelinedCore(torch.autograd.Function):
    """"a2a" mode, fused kernel available: the per-slice all-to-alls run on a side stream while
    the fused P2+P3 kernel works through the slices that have already arrived (forward), and
    the slices the backward kernel has finished leave while it works on the next ones.
    One slice of S4 is ~6 ms of gather and ~2 GB of exchange, so after the first slice the
    exchange is hidden behind compute.  Consecutive one-slice launches alternate between two
    compute streams (`compute_lanes`): the persistent kernel of slice k+1 becomes resident block by
    block as the blocks of slice k run out of tiles, so the tail of one launch overlaps the head of
    the next instead of draining the chip sixteen times per pass (measured at world size 1:
    DESIGN.md §6).  On CPU tensors (gloo tests) the same code runs without streams."""

    @staticmethod
    def forward(ctx, send, W, layer, act):
        G, Tl, N = layer.G, layer.Tl, layer.N
        Nl, F = send.shape[-2], send.shape[-1]
        K = ops.kernels
        per_slice_w = W.dim() == 3
        Nf = W.shape[-1]
        dev = send.device
        use_streams = send.is_cuda
        need_w = ctx.needs_input_grad[1]
        act_on = bool(act) and act != "none"
        Xt = torch.empty(Tl, N, F, dtype=send.dtype, device=dev)
        Y = torch.empty(Tl, N, Nf, dtype=send.dtype, device=dev)
        AX = torch.empty(Tl, N, F, dtype=send.dtype, device=dev) if need_w else None
        pre = torch.empty(Tl, N, Nf, dtype=send.dtype, device=dev) if act_on else None
        if use_streams:
            main = torch.cuda.current_stream(dev)
            comm = layer.comm_stream()
            comm.wait_stream(main)  # the send buffer (P1 output) is complete
            evs = []
            with torch.cuda.stream(comm):
                for kk in range(Tl):
                    dist.all_to_all_single(Xt[kk].view(G, Nl, F), send[kk], group=layer.group)
                    ev = torch.cuda.Event()
                    ev.record(comm)
                    evs.append(ev)
            lanes = layer.compute_lanes(main)   # consecutive one-slice launches alternate between two streams
            for s2 in lanes:
                if s2 is not main:
                    s2.wait_stream(main)        # W and the output allocations are ordered before the first launch there
        for kk in range(Tl):
            Wk = W[kk:kk + 1] if per_slice_w else W
            outs = (Y[kk:kk + 1], AX[kk:kk + 1] if need_w else None, pre[kk:kk + 1] if act_on else None)
            if use_streams:
                lane = lanes[kk % len(lanes)]
                lane.wait_event(evs[kk])
                with torch.cuda.stream(lane):
                    K.spmm_gemm(layer.A_views[kk], Xt[kk:kk + 1], Wk, act=act, want_ax=need_w, want_pre=act_on, out=outs,
                                grid_reserve=layer.grid_reserve)
            else:
                dist.all_to_all_single(Xt[kk].view(G, Nl, F), send[kk], group=layer.group)
                K.spmm_gemm(layer.A_views[kk], Xt[kk:kk + 1], Wk, act=act, want_ax=need_w, want_pre=act_on, out=outs,
                            grid_reserve=layer.grid_reserve)
        if use_streams:
            for s2 in lanes:
                if s2 is not main:
                    main.wait_stream(s2)
        ctx.layer, ctx.act, ctx.shape = layer, (act if act_on else None), (Tl, G, Nl, F)
        empty = torch.empty(0, device=dev)
        ctx.save_for_backward(W, AX if need_w else empty, pre if act_on else empty)
        return Y

    @staticmethod
    def backward(ctx, dY):
        layer = ctx.layer
        W, AX, pre = ctx.saved_tensors
        Tl, G, Nl, F = ctx.shape
        K = ops.kernels
        dY = dY.contiguous()
        if ctx.act is not None:
            dY = K.act_bwd(pre, dY, ctx.act)
        per_slice_w = W.dim() == 3
        dsend = dW = None
        if ctx.needs_input_grad[0]:
            dev = dY.device
            use_streams = dY.is_cuda
            dXt = torch.empty(Tl, layer.N, F, dtype=dY.dtype, device=dev)
            dsend = torch.empty(Tl, G, Nl, F, dtype=dY.dtype, device=dev)
            if use_streams:
                main = torch.cuda.current_stream(dev)
                comm = layer.comm_stream()
                comm.wait_stream(main)  # dsend/dXt allocations and earlier work are ordered before the exchange
                lanes = layer.compute_lanes(main)
                for s2 in lanes:
                    if s2 is not main:
                        s2.wait_stream(main)    # dY (and its activation gradient) is complete
            for kk in range(Tl):
                Wk = W[kk:kk + 1] if per_slice_w else W
                if use_streams:
                    lane = lanes[kk % len(lanes)]
                    with torch.cuda.stream(lane):
                        K.spmm_gemm(layer.At_views[kk], dY[kk:kk + 1], Wk, trans_w=True, tag="spmm_gemm_T",
                                    out=(dXt[kk:kk + 1], None, None), grid_reserve=layer.grid_reserve)
                    ev = torch.cuda.Event()
                    ev.record(lane)
                    comm.wait_event(ev)
                    with torch.cuda.stream(comm):
                        dist.all_to_all_single(dsend[kk], dXt[kk].view(G, Nl, F), group=layer.group)
                else:
                    K.spmm_gemm(layer.At_views[kk], dY[kk:kk + 1], Wk, trans_w=True, tag="spmm_gemm_T",
                                out=(dXt[kk:kk + 1], None, None), grid_reserve=layer.grid_reserve)
                    dist.all_to_all_single(dsend[kk], dXt[kk].view(G, Nl, F), group=layer.group)
            if use_streams:
                for s2 in lanes:
                    if s2 is not main:
                        main.wait_stream(s2)
        if ctx.needs_input_grad[1]:
            # dW does not depend on the exchange: it runs while the last slices are still leaving
            dW = K.gemm_dw(AX, dY, per_slice=per_slice_w)
        if ctx.needs_input_grad[0] and dY.is_cuda:
            torch.cuda.current_stream(dY.device).wait_stream(layer.comm_stream())
        return dsend, dW, None, None


# ---------------------------------------------------------------------------------------
# the sharded layer
# ---------------------------------------------------------------------------------------
class ShardedTMGCNLayer:
    """Y = act((Â ⋆ (M ×₁ X)) · W) with the frontal slices sharded over `group`.

    A_local : BatchedCSR of this rank's slices [k0,k1) (T_local x N x N)
    M       : full [T,T] mixing matrix (replicated, tiny)
    exchange: "a2a" (X node-sharded [T, N/G, F]) or "allgather" (X slice-sharded [T/G, N, F])
    Output is slice-sharded [T/G, N, F1] in both modes.  World size 1 needs no process group.
    Even splits are required (T % G == 0, and N % G == 0 for "a2a").
    """

    def __init__(self, A_local: BatchedCSR, M, T: int, group=None, exchange: str = "a2a",
                 apply_m: bool = True, fuse: Optional[bool] = None, pipeline: bool = True,
                 force_collectives: bool = False, local_only: bool = False, grid_reserve: Optional[int] = None,
                 gather_chunk_nodes: Optional[int] = None, cu_reserve: Optional[int] = None):
        # local_only: ignore any initialised process group (an unsharded layer inside a
        # distributed job, e.g. to cross-check a sharded result)
        self.rank, self.G = (0, 1) if local_only else _world(group)
        # force_collectives: run the exchange code at world size 1 too (exercises the RCCL path on
        # a single GPU; needs an initialised process group)
        self.collective = self.G > 1 or (force_collectives and dist.is_initialized())
        self.pipeline = pipeline
        # RCCL's exchange kernels run on the side stream while the persistent fused kernel works; they
        # can only become resident if that kernel does not hold the whole chip.  Two mechanisms:
        #   cu_reserve    the one-slice launches go to CU-MASKED streams that leave that many CUs
        #                 alone (hipExtStreamCreateWithCUMask).  Default 32 whenever a real exchange
        #                 runs beside the kernel: measured at world size 1, 16 / 32 CUs cost the gather
        #                 nothing measurable (209.9 / 212.1 vs 211.6 ms per step), 64 cost 7.5 %.
        #   grid_reserve  block slots the persistent grid leaves free (a per-launch argument of the
        #                 C-ABI).  Round 2's default of one slot per CU is now 0: an RCCL workgroup
        #                 (256 threads, 132 VGPRs, 20 KB LDS: rocprofv3) does not fit beside three
        #                 128-VGPR blocks of the fused kernel on a CU (3 x 128 + 136 > 512), so the
        #                 free slot could not host it, and it cost 8 % at world size 1
        #                 (profiles/archive/r3i_cu_mask_rccl_world1.txt).
        real_exchange = self.G > 1 and pipeline and exchange == "a2a"
        if grid_reserve is None:
            grid_reserve = 0
        self.grid_reserve = int(grid_reserve)   # passed with every fused launch of THIS layer; nothing process-wide
        if cu_reserve is None:
            cu_reserve = int(os.environ.get("TMGCN_CU_RESERVE", "32" if real_exchange else "0"))
        self.cu_reserve = int(cu_reserve)
        # "allgather" mode: nodes per chunk of the chunked gather (None: sized so that one
        # [T, Nc, F] chunk buffer stays near GATHER_CHUNK_BYTES; 0: the unchunked literal form)
        self.gather_chunk_nodes = gather_chunk_nodes
        self._gbufs = None
        self._lanes = None
        self._lane2 = None
        self.pipeline_lanes = int(os.environ.get("TMGCN_PIPELINE_LANES", "2"))   # A/B knob; 2 = alternate two compute streams
        self._comm_stream = None
        self._views = None
        self.group = group
        if exchange not in ("a2a", "allgather"):
            raise RuntimeError(f"unknown exchange {exchange!r}")
        if T % self.G:
            raise RuntimeError(f"T={T} is not divisible by the world size {self.G}")
        self.exchange = exchange
        self.T, self.Tl, self.N = T, T // self.G, A_local.N
        if A_local.T != self.Tl:
            raise RuntimeError(f"rank {self.rank}: adjacency shard has {A_local.T} slices, expected {self.Tl}")
        if exchange == "a2a" and self.N % self.G:
            raise RuntimeError(f"N={self.N} is not divisible by the world size {self.G}")
        self.k0 = self.rank * self.Tl
        self.A = A_local
        self.apply_m = apply_m
        self.fuse = fuse  # None: fused P2+P3 kernel whenever it supports the widths
        self.Mop = ops.MOperator(M, A_local.device) if apply_m else None
        if apply_m and self.Mop.T != T:
            raise RuntimeError(f"M is {self.Mop.T}x{self.Mop.T}, expected {T}x{T}")

    def gather_chunks(self, F: Optional[int] = None) -> List[Tuple[int, int]]:
        """Node ranges of the chunked all-gather.  The automatic size needs F (bytes per node of a
        [T, Nc, F] buffer); once chosen it is kept, so forward and backward agree."""
        if self.gather_chunk_nodes is None:
            if F is None:
                raise RuntimeError("gather_chunks: the chunk size has not been chosen yet")
            self.gather_chunk_nodes = chunk_nodes_for(self.T, self.N, F)
        nc = max(1, min(self.N, int(self.gather_chunk_nodes)))
        return [(c0, min(self.N, c0 + nc)) for c0 in range(0, self.N, nc)]

    def gather_buffers(self, F: int, dtype, device):
        """The two alternating [T, Nc, F] chunk buffers (flat), allocated once per layer: they are
        written on the side stream and read on the main one, so they never go back to the
        allocator between passes."""
        nc = self.gather_chunks(F)[0][1]
        need = self.T * nc * F
        if self._gbufs is None or self._gbufs[0].numel() < need or self._gbufs[0].dtype != dtype or \
                self._gbufs[0].device != torch.device(device):
            self._gbufs = [torch.empty(need, dtype=dtype, device=device) for _ in range(2)]
        return self._gbufs

    def compute_lanes(self, main):
        """Streams the one-slice launches of the pipelined path alternate between: the caller's
        stream and one more (`pipeline_lanes` = 1 keeps everything on the caller's stream)."""
        if self.cu_reserve > 0:              # CU-masked streams (never the caller's own stream)
            if self._lanes is None:
                try:
                    self._lanes = [cu_masked_stream(self.A.device, self.cu_reserve, lane) for lane in range(max(1, self.pipeline_lanes))]
                except (RuntimeError, OSError, AttributeError) as e:
                    # a performance device, not a correctness one: without it the exchange merely overlaps
                    # less.  The choice is local to this rank (no collective depends on it).
                    import warnings
                    warnings.warn(f"tmgcn_amd: CU-masked streams are not available ({e}); the pipelined exchange runs on "
                                  "ordinary streams", RuntimeWarning)
                    self.cu_reserve = 0
                    return self.compute_lanes(main)
            return self._lanes
        if self.pipeline_lanes <= 1:
            return [main]
        if self._lane2 is None:
            self._lane2 = torch.cuda.Stream(device=self.A.device)
        return [main, self._lane2]

    def comm_stream(self):
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.A.device)
        return self._comm_stream

    @property
    def A_views(self):
        if self._views is None:
            self._views = (self.A.slice_views(), self.A.transpose().slice_views())
        return self._views[0]

    @property
    def At_views(self):
        self.A_views
        return self._views[1]

    def input_shape(self, F: int):
        if not self.collective or self.exchange == "allgather":
            return (self.Tl, self.N, F)
        return (self.T, self.N // self.G, F)

    def __call__(self, X: torch.Tensor, W: torch.Tensor, act=None) -> torch.Tensor:
        if tuple(X.shape[:2]) != self.input_shape(X.shape[2])[:2]:
            raise RuntimeError(f"rank {self.rank}: input {tuple(X.shape)} does not match the "
                               f"'{self.exchange}' layout {self.input_shape(X.shape[2])}")
        shared_w = W.dim() == 2
        if not self.collective:
            Xt = ops.m_transform(X, self.Mop) if self.apply_m else X
        elif self.exchange == "a2a":
            Nl, F = X.shape[1], X.shape[2]
            if self.apply_m:
                # P1 on whole tube fibres, written straight into the send layout [Tl][G][Nl][F]
                send = ops.m_transform(X, self.Mop, y_group_rows=self.Tl)
            else:
                send = X.view(self.G, self.Tl, Nl, F).transpose(0, 1).contiguous()
            send = send.view(self.Tl, self.G, Nl, F)
            can_fuse = hasattr(ops.kernels, "spmm_gemm_supported") and \
                ops.kernels.spmm_gemm_supported(F, W.shape[-1]) and ops.kernels.spmm_gemm_supported(W.shape[-1], F)
            if self.pipeline and can_fuse and self.fuse is not False:
                if shared_w:
                    W = _SharedWeight.apply(W, self.group)
                return _PipelinedCore.apply(send, W, self, act)
            Xt = _NodeToSlice.apply(send, self.Tl, self.N, self.group)
        elif not self.apply_m:
            Xt = X                                   # no M: a rank's own slices are all it needs — no exchange
        elif self.gather_chunk_nodes == 0:
            # the literal form: the whole [T,N,F] tensor on every rank (fits only for small N·F or G)
            Xf = _AllGatherSlices.apply(X, self.group)
            # own output slices only: rows [k0, k0+Tl) of M against all T input slices
            Xt = ops.m_transform(Xf, self.Mop, row_off=self.k0, col_off=0, T_out=self.Tl)
        else:
            self.gather_chunks(X.shape[2])           # fixes the chunk size on first use
            Xt = _ChunkedGatherTransform.apply(X, self)
        if shared_w and self.collective:
            W = _SharedWeight.apply(W, self.group)      # condensed_W: dW summed over ranks
        return ops.spmm_feature_gemm(self.A, Xt, W, act=act, fuse=self.fuse)

    def to_node_sharded(self, Y: torch.Tensor) -> torch.Tensor:
        """Slice-sharded [T/G, N, F] -> node-sharded [T, N/G, F] (input of a following "a2a" layer)."""
        if not self.collective:
            return Y
        Nl, F = self.N // self.G, Y.shape[2]
        recv = _SliceToNode.apply(Y, self.group)          # [Tl][G][Nl][F], block kk = slice kk of every rank
        return recv.transpose(0, 1).reshape(self.T, Nl, F)


# ---------------------------------------------------------------------------------------
# slice sharding of the drop-in models (layers.EmbeddingGCN / EmbeddingGCN2 / EmbeddingKWGCN, group=…)
# ---------------------------------------------------------------------------------------
class _GatherSlices(torch.autograd.Function):
    """Slice-sharded [Tl_r, …] -> replicated [T, …] (ranks may own one slice more or less: shards
    are padded to the largest).  Each rank goes on to compute something different from the
    gathered tensor (its own rows of M ×₁ ·), so the adjoint is a reduce-scatter."""

    @staticmethod
    def forward(ctx, x, shard):
        ctx.shard = shard
        G, Tmax = shard.G, shard.Tmax
        tail = tuple(x.shape[1:])
        pad = x.new_zeros((Tmax,) + tail)
        pad[:x.shape[0]] = x
        out = x.new_empty((G * Tmax,) + tail)
        dist.all_gather_into_tensor(out, pad, group=shard.group)
        out = out.view((G, Tmax) + tail)
        return torch.cat([out[r, :b - a] for r, (a, b) in enumerate(shard.bounds)], dim=0)

    @staticmethod
    def backward(ctx, d):
        shard = ctx.shard
        G, Tmax = shard.G, shard.Tmax
        tail = tuple(d.shape[1:])
        pad = d.new_zeros((G, Tmax) + tail)
        for r, (a, b) in enumerate(shard.bounds):
            pad[r, :b - a] = d[a:b]
        out = d.new_empty((Tmax,) + tail)
        dist.reduce_scatter_tensor(out, pad.view((G * Tmax,) + tail), op=dist.ReduceOp.SUM, group=shard.group)
        return out[:shard.Tl].contiguous(), None


class _GatherRows(torch.autograd.Function):
    """Per-rank rows (the logits of the edges a rank owns) -> the full [E, C] result in the caller's
    edge order, on every rank.  Every rank then evaluates the SAME loss on it, so the gradient of
    the local rows is simply their rows of the upstream gradient (no sum over ranks: the loss is
    one replicated scalar, not G different ones)."""

    @staticmethod
    def forward(ctx, local, shard, counts, gather_index, mine):
        ctx.mine = mine
        G, Emax = shard.G, max(max(counts), 1)
        tail = tuple(local.shape[1:])
        pad = local.new_zeros((Emax,) + tail)
        pad[:local.shape[0]] = local
        out = local.new_empty((G * Emax,) + tail)
        dist.all_gather_into_tensor(out, pad, group=shard.group)
        return out[gather_index]

    @staticmethod
    def backward(ctx, d):
        return d[ctx.mine].contiguous(), None, None, None, None


class SliceShard:
    """Slice ownership of one rank inside `group`, and the three collectives a slice-sharded
    drop-in model needs (all autograd-aware):
      * m_transform  — the ONE exchange of a layer that mixes slices (apply_M_twice / use_Minv
                       branches, ehf:224, 332, 341-346): all-gather of the slice-sharded activation,
                       then only this rank's rows of M (or M⁻¹);
      * shared       — replicated parameters: gradients summed over the ranks (all-reduce);
      * gather_rows  — per-rank edge logits -> the full [E, C] tensor in the caller's edge order.
    The reference's as-run models (default layer-2 branch) need no activation exchange at all."""

    def __init__(self, group, T: int):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("group= needs an initialised torch.distributed process group")
        self.group = group
        self.rank, self.G = dist.get_rank(group), dist.get_world_size(group)
        self.T = int(T)
        self.bounds = even_bounds(self.T, self.G)
        self.k0, self.k1 = self.bounds[self.rank]
        self.Tl = self.k1 - self.k0
        self.Tmax = max(b - a for a, b in self.bounds)

    def local(self, seq):
        """This rank's slices of a per-slice sequence (list of adjacency slices, or a [T, …] tensor)."""
        return seq[self.k0:self.k1]

    def gather_slices(self, x_local: torch.Tensor) -> torch.Tensor:
        return _GatherSlices.apply(x_local.contiguous(), self)

    def m_transform(self, x_local: torch.Tensor, op: "ops.MOperator") -> torch.Tensor:
        full = self.gather_slices(x_local)
        return ops.m_transform(full, op, row_off=self.k0, col_off=0, T_out=self.Tl)

    def shared(self, w: torch.Tensor) -> torch.Tensor:
        return _SharedWeight.apply(w, self.group)

    def edge_index(self, edges: torch.Tensor, N: int, device):
        """(EdgeIndex of the edges whose slice this rank owns — slice numbers re-based to the shard —,
        per-rank counts, gather_index [E] into the rank-major padded gather, positions of this rank's edges)."""
        e = edges.detach()
        ops.EdgeIndex(e, N, "cpu" if e.device.type == "cpu" else e.device, T=self.T)     # validate the full set once
        t = e[0]
        owner = torch.zeros_like(t)
        for r, (a, b) in enumerate(self.bounds):
            owner[(t >= a) & (t < b)] = r
        counts = [int((owner == r).sum()) for r in range(self.G)]
        Emax = max(max(counts), 1)
        within = torch.zeros_like(t)
        for r in range(self.G):
            m = owner == r
            within[m] = torch.arange(counts[r], dtype=t.dtype, device=t.device)
        gather_index = (owner * Emax + within).to(device)
        mine = torch.nonzero(owner == self.rank).reshape(-1)
        e_loc = e[:, mine].clone()
        e_loc[0] -= self.k0
        return ops.EdgeIndex(e_loc, N, device, T=max(self.Tl, 1)), counts, gather_index, mine.to(device)

    def gather_rows(self, local: torch.Tensor, counts, gather_index, mine) -> torch.Tensor:
        return _GatherRows.apply(local.contiguous(), self, counts, gather_index, mine)
