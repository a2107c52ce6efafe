"""GPU: the reference's REAL operand through the WIDE kernels (VERDICT r5 missing 3).

Every F >= 16 measurement and parity case of rounds 1-5 used rows of at least 7 entries and uniformly random columns.  The
one operand the reference ships — Ât of its chess data, func_MProduct over the symmetrised, windowed slices
(read_data.py:116-127, 204-223): N = 7 301, T = 80, 2.32 M entries, two rows of three holding the self loop only — had only
met the F = 2 -> 6 -> 6 kernels.  Here the device-built Ât of fixture G10 (bit-exact pattern against the reference's own
output: test_gpu_g10_chess_full.py) carries seeded F = 128 / 64 / 16 features through

    tmgcn_spmm_csr_batched_f32           vs ref_spmm                       (oracle/tmgcn_ref.c)
    tmgcn_spmm_gemm_f32 (+ AX)           vs ref_gemm(ref_spmm(...))
    the same on the transposed CSR, Wᵀ   vs ref_gemm(ref_spmm(Âᵀ, dY), Wᵀ)  (the backward pair, ehf:206-207 + 222 under autograd)
    ops.spmm_feature_gemm + backward     dX, dW vs the oracle

at 1e-5 · max|ref|.  84 % of this operand's 64-row tiles (53 % of its entries) hold at most 512 entries and take the
entry-major walk of csrc/spmm_row.h ("Short tiles"); the tiles of the late, dense slices take the row-per-wave path (rows of
up to 83 entries), so one launch mixes both and the fused kernel's SpMM intermediate must still equal the plain kernel's
bit for bit."""
import numpy as np
import pytest
import torch

from _g10 import G10
from _util import REL_TOL, assert_close, cptr, load_c_oracle
from tmgcn_amd import adjacency, ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def operand():
    g = G10()
    k, i, j = g.raw
    Chat, _ = adjacency.build_adjacency(k, i, j, np.ones(len(k), np.float32), g.TT, g.N, M=None, window=10)
    A = adjacency.m_product_csr(Chat.slices(0, g.T), g.M)
    rk, ri, rj, rv = g.Ct()                                 # the reference's own Ct_train, entry for entry
    assert A.nnz == len(rv) and np.array_equal(A.col.cpu().numpy().astype(np.int64), rj)
    return A, A.to("cpu"), A.transpose(), A.transpose().to("cpu")


def _ref_spmm(csr, X):
    Y = torch.empty_like(X)
    load_c_oracle().ref_spmm(cptr(csr.rowptr), cptr(csr.col), cptr(csr.val), cptr(X), cptr(Y), csr.n_rows, csr.N, X.shape[2])
    return Y


def _ref_gemm(A, W, trans_w):
    T, N, K = A.shape
    Nf = W.shape[0] if trans_w else W.shape[1]
    Y = torch.empty(T, N, Nf)
    load_c_oracle().ref_gemm(cptr(A), cptr(W), cptr(Y), T * N, K, Nf, int(trans_w), 0, 0)
    return Y


def test_structure_is_what_the_short_tile_path_is_for(operand):
    A, Ac, _, _ = operand
    cnt = (Ac.rowptr[1:] - Ac.rowptr[:-1]).view(Ac.T, Ac.N)
    assert int(cnt.median()) == 1 and float((cnt == 1).sum()) / cnt.numel() > 0.6 and 3.5 < Ac.nnz / cnt.numel() < 4.5
    per = (Ac.N + 63) // 64
    pad = torch.zeros(Ac.T, per * 64, dtype=torch.int64)
    pad[:, :Ac.N] = cnt
    tiles = pad.view(Ac.T, per, 64).sum(-1)                 # entries per tile: tiles restart at every slice (spmm_row.h)
    short = float((tiles <= 512).sum()) / tiles.numel()
    assert 0.8 < short < 1.0, short                         # both paths in one launch
    assert int(cnt.max()) > 64                              # and rows of several 64-entry batches on the row-per-wave path


@pytest.mark.parametrize("F,Nf", [(128, 128), (64, 128), (16, 48), (120, 64)])
def test_real_operand_plain_and_fused_forward(operand, F, Nf):
    A, Ac, _, _ = operand
    g = torch.Generator().manual_seed(F * 13 + Nf)
    X = torch.randn(A.T, A.N, F, generator=g)
    W = torch.randn(F, Nf, generator=g) * 0.2
    ref_ax = _ref_spmm(Ac, X)
    K = ops.kernels
    Y1 = K.spmm(A, X.to(DEV))
    assert_close(Y1, ref_ax, REL_TOL, f"chess Ât, plain SpMM F={F}")
    assert torch.equal(Y1, K.spmm(A, X.to(DEV)))
    Y, AX, _ = K.spmm_gemm(A, X.to(DEV), W.to(DEV), want_ax=True)
    assert_close(AX, ref_ax, REL_TOL, f"chess Ât, fused SpMM intermediate F={F}")
    assert_close(Y, _ref_gemm(ref_ax, W, False), REL_TOL, f"chess Ât, fused SpMM+GEMM {F}->{Nf}")
    assert torch.equal(AX, Y1), "fused and plain kernels sum a row in different orders"
    Y2, _, _ = K.spmm_gemm(A, X.to(DEV), W.to(DEV))
    assert torch.equal(Y, Y2)


@pytest.mark.parametrize("F,Nf", [(128, 128), (64, 32)])
def test_real_operand_backward_pair_on_the_transposed_csr(operand, F, Nf):
    """dXt = Âᵀ(dY·Wᵀ) = (Âᵀ·dY)·Wᵀ: the fused kernel on the transposed CSR with Wᵀ (trans_w), dY of width Nf."""
    A, _, At, Atc = operand
    g = torch.Generator().manual_seed(F + 7 * Nf)
    dY = torch.randn(A.T, A.N, Nf, generator=g)
    W = torch.randn(F, Nf, generator=g) * 0.2
    ref = _ref_gemm(_ref_spmm(Atc, dY), W, True)
    if not ops.kernels.spmm_gemm_supported(Nf, F):
        pytest.skip("width pair outside the fused kernel")
    dX, _, _ = ops.kernels.spmm_gemm(At, dY.to(DEV), W.to(DEV), trans_w=True)
    assert_close(dX, ref, REL_TOL, f"chess Âᵀ, fused backward pair {Nf}->{F}")


def test_real_operand_layer_autograd(operand):
    """The differentiable operator the layers call (ops.spmm_feature_gemm) on the real operand at F = 128 -> 128:
    Y, dX and dW = AXᵀ·dY against the oracle."""
    A, Ac, _, Atc = operand
    F = 128
    g = torch.Generator().manual_seed(5)
    X = torch.randn(A.T, A.N, F, generator=g)
    W = torch.randn(F, F, generator=g) * 0.1
    dY = torch.randn(A.T, A.N, F, generator=g)
    Xd, Wd = X.to(DEV).requires_grad_(True), W.to(DEV).requires_grad_(True)
    Y = ops.spmm_feature_gemm(A, Xd, Wd)
    Y.backward(dY.to(DEV))
    ref_ax = _ref_spmm(Ac, X)
    assert_close(Y, _ref_gemm(ref_ax, W, False), REL_TOL, "layer Y")
    assert_close(Xd.grad, _ref_gemm(_ref_spmm(Atc, dY), W, True), REL_TOL, "layer dX")
    dW = torch.empty(F, F)
    load_c_oracle().ref_gemm_dw(cptr(ref_ax), cptr(dY), cptr(dW), A.T * A.N, F, F, 0)
    assert_close(Wd.grad, dW, REL_TOL, "layer dW")


def test_block_diagonal_replication_keeps_rows_and_results(operand):
    """synth.tile_block_diagonal (bench.py's `roofline_real_structure` leg): three copies of slices 79 and 4 on the block
    diagonal — row lengths and values repeat, and the product on replicated features is the original product, replica by
    replica (tile boundaries move with the replica's offset, so paths may differ: tolerance, not bits)."""
    A, Ac, _, _ = operand
    B = synth.tile_block_diagonal(A, 3, [79, 4])
    assert B.T == 2 and B.N == 3 * A.N
    c0 = (A.rowptr[1:] - A.rowptr[:-1]).view(A.T, A.N)
    c1 = (B.rowptr[1:] - B.rowptr[:-1]).view(2, 3, A.N)
    assert torch.equal(c1[0], c0[79].expand(3, -1)) and torch.equal(c1[1], c0[4].expand(3, -1))
    F = 128
    X = torch.randn(A.T, A.N, F, generator=torch.Generator().manual_seed(9)).to(DEV)
    Y = ops.kernels.spmm(A, X)
    Xb = torch.stack([X[79].repeat(3, 1), X[4].repeat(3, 1)])
    Yb = ops.kernels.spmm(B, Xb).view(2, 3, A.N, F)
    for q in range(3):
        assert_close(Yb[0, q], Y[79], REL_TOL, f"replica {q} of slice 79")
        assert_close(Yb[1, q], Y[4], REL_TOL, f"replica {q} of slice 4")
