"""GPU, maximum sizes: a batched CSR with more than 2^31 stored non-zeros (40 slices of the S4 graph,
2.64 G entries, 21 GB) — the offsets are int64 end to end.  Size-independent properties only; needs
~155 GB of HBM (skipped on smaller devices)."""
import pytest
import torch

from tmgcn_amd import ops, synth

pytestmark = pytest.mark.gpu


def test_more_than_2_31_nonzeros():
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs a 288 GB device")
    dev = "cuda"
    T, N, F, deg = 40, 2_000_000, 16, 32
    A = synth.device_er_csr(T, N, deg, dev)
    assert A.nnz == T * N * (deg + 1) and A.nnz > 2 ** 31
    K = ops.kernels
    Y = K.spmm(A, torch.ones(T, N, F, device=dev))          # row-normalised: Â·1 = 1 in every slice
    assert abs(float(Y.min()) - 1.0) < 1e-6 and abs(float(Y.max()) - 1.0) < 1e-6
    del Y
    X = torch.rand(T, N, F, device=dev)
    Yfull = K.spmm(A, X)
    # the last slice lives entirely beyond offset 2^31: it must equal the same slice built on its own
    A_last = synth.device_er_csr(1, N, deg, dev, first_slice=T - 1)
    assert torch.equal(Yfull[T - 1:], K.spmm(A_last, X[T - 1:].contiguous()))
    W = torch.randn(F, 8, device=dev)
    Yf, _, _ = K.spmm_gemm(A, X, W)                          # fused kernel, dynamic tile counter over 1.25 M tiles
    ref = Yfull.reshape(-1, F) @ W
    assert float((Yf.reshape(-1, 8) - ref).abs().max() / ref.abs().max()) < 1e-5
    del Yf, ref
    At = A.transpose()                                        # native rocPRIM sort of 2.64 G keys
    Yb = torch.rand(T, N, F, device=dev)
    Z = K.spmm(At, Yb)
    lhs = float((Yfull[T - 2:].double() * Yb[T - 2:].double()).sum())
    rhs = float((X[T - 2:].double() * Z[T - 2:].double()).sum())
    assert abs(lhs - rhs) <= 1e-9 * abs(lhs)                  # <Âx, y> = <x, Âᵀy> on the slices beyond 2^31


def test_mproduct_merge_with_more_than_2_31_output_entries():
    """The segmented-merge M-product where its OUTPUT exceeds 2^31 stored entries (T = 24 slices of a
    1 M-node graph, 8+1 per row, 20 diagonals: 2.7 G entries, 22 GB — the expand + sort form would need
    > 100 GB of keys here): int64 offsets end to end; row sums follow from the inputs' (every input
    row sums to 1, so output row (k, r) sums to the sum of M[k, :] over the band); the last slice —
    wholly beyond offset 2^31 — equals the same product formed on the last 20 input slices alone."""
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs a 288 GB device")
    from tmgcn_amd import adjacency
    dev = "cuda"
    T, N, deg, b = 24, 1_000_000, 8, 20
    A = synth.device_er_csr(T, N, deg, dev)
    M = synth.band_M(T, b, "matlab")
    out = adjacency.m_product_csr(A, M, algo="merge")
    assert out.nnz > 2 ** 31 and int(out.rowptr[-1]) == out.nnz
    assert bool((out.rowptr[1:] >= out.rowptr[:-1]).all())
    ones = torch.ones(T, N, 1, device=dev)
    rowsum = ops.kernels.spmm(out, ones)[..., 0]                        # [T, N]
    want = torch.from_numpy(M.sum(1)).float().to(dev)[:, None].expand(T, N)
    assert float((rowsum - want).abs().max()) < 1e-5
    # last output slice from the slices it depends on only: same entries, same bits
    j0 = T - b
    tail = adjacency.m_product_csr(A.slices(j0, T), M[j0:, j0:], algo="merge")
    lo, hi = int(out.rowptr[(T - 1) * N]), int(out.rowptr[T * N])
    lo2, hi2 = int(tail.rowptr[(b - 1) * N]), int(tail.rowptr[b * N])
    assert hi - lo == hi2 - lo2 and lo > 2 ** 31
    assert torch.equal(out.col[lo:hi], tail.col[lo2:hi2]) and torch.equal(out.val[lo:hi], tail.val[lo2:hi2])
