"""GPU: every HIP kernel, called through the C-ABI, against the CPU oracle on seeded inputs.

Tolerance: max|Δ| <= 1e-5 · max|ref| (fp32 kernels vs the oracle's fp64-accumulate / fp32-store;
SURVEY §8c).  Integer/index work (CSR build) is bit-exact.
"""
import numpy as np
import pytest
import torch

from _util import max_rel_err, REL_TOL, assert_close, cptr, load_c_oracle
from tmgcn_amd import ops, synth
from tmgcn_amd.csr import BatchedCSR

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rand_csr(T, N, avg_deg, seed, empty_rows=True):
    g = torch.Generator().manual_seed(seed)
    nnz = int(T * N * avg_deg)
    k = torch.randint(0, T, (nnz,), generator=g)
    i = torch.randint(0, N, (nnz,), generator=g)
    j = torch.randint(0, N, (nnz,), generator=g)
    if empty_rows and N > 3:
        keep = i != 1  # row 1 of every slice stays empty
        k, i, j = k[keep], i[keep], j[keep]
    v = torch.randn(k.numel(), generator=g, dtype=torch.float64)
    return BatchedCSR.from_coo(k, i, j, v, T, N)


def ref_spmm(csr, X):
    lib = load_c_oracle()
    Y = torch.empty_like(X)
    lib.ref_spmm(cptr(csr.rowptr), cptr(csr.col), cptr(csr.val), cptr(X), cptr(Y), csr.n_rows, csr.N, X.shape[2])
    return Y


# ------------------------------------------------------------------------------------- P2
@pytest.mark.parametrize("F", [1, 2, 3, 4, 6, 8, 5, 12, 16, 20, 32, 64, 100, 128, 256, 260, 512, 1000])
@pytest.mark.parametrize("T,N,deg", [(3, 50, 4.0), (2, 301, 40.0), (1, 7, 1.5)])
def test_spmm_vs_oracle(F, T, N, deg):
    csr = rand_csr(T, N, deg, seed=F * 7 + N)
    X = torch.randn(T, N, F, generator=torch.Generator().manual_seed(1))
    Y = ops.kernels.spmm(csr.to(DEV), X.to(DEV))
    assert_close(Y, ref_spmm(csr, X), REL_TOL, f"spmm F={F}")


def test_spmm_long_rows_and_bitwise_reproducible():
    # one hub row with 5000 non-zeros (several 64-wide batches) next to empty rows
    T, N, F = 2, 400, 128
    k = torch.cat([torch.zeros(5000, dtype=torch.long), torch.ones(10, dtype=torch.long)])
    i = torch.cat([torch.full((5000,), 3), torch.arange(10)])
    j = torch.randint(0, N, (5010,), generator=torch.Generator().manual_seed(5))
    v = torch.randn(5010, generator=torch.Generator().manual_seed(6))
    csr = BatchedCSR.from_coo(k, i, j, v, T, N)
    X = torch.randn(T, N, F, generator=torch.Generator().manual_seed(7))
    Y1 = ops.kernels.spmm(csr.to(DEV), X.to(DEV))
    Y2 = ops.kernels.spmm(csr.to(DEV), X.to(DEV))
    assert torch.equal(Y1, Y2)
    assert_close(Y1, ref_spmm(csr, X), REL_TOL, "hub row")


def _csr_with_row_lengths(T, N, lengths, seed):
    """CPU BatchedCSR with the given row lengths ({(slice, row): entries}, every other row 3 entries), random
    columns (duplicates kept: they add up like uncoalesced COO entries in sparse.mm) and values."""
    g = torch.Generator().manual_seed(seed)
    cnt = torch.full((T * N,), 3, dtype=torch.int64)
    for (k, i), n in lengths.items():
        cnt[k * N + i] = n
    rowptr = torch.zeros(T * N + 1, dtype=torch.int64)
    torch.cumsum(cnt, 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    col = torch.randint(0, N, (nnz,), generator=g, dtype=torch.int32)
    val = torch.randn(nnz, generator=g) / cnt.repeat_interleave(cnt).float().sqrt()
    return BatchedCSR(rowptr, col, val, T, N)


LONG_ROW_CASES = {
    # rows at and around the split threshold (256 entries, csrc/spmm_row.h) and its 64-entry quarters, several long rows in
    # one tile, long rows in the last (ragged) tile of a slice, an empty row, and 10^4 / 10^5-entry hubs (tiles the blocks
    # take first: > 8192 entries)
    "threshold": (3, 1000, {(0, 0): 256, (0, 1): 257, (0, 2): 258, (0, 3): 320, (0, 64): 511, (0, 65): 512, (0, 66): 513,
                            (1, 5): 1025, (1, 6): 0, (1, 7): 255, (2, 999): 700, (2, 998): 300, (2, 960): 4097, (1, 999): 257}),
    "hubs": (2, 100_003, {(0, 17): 100_000, (0, 18): 10_000, (0, 4000): 300, (1, 100_002): 100_000, (1, 50_000): 65_536,
                          (1, 50_001): 9_000, (1, 0): 20_000, (0, 99_968): 12_345}),
    "one heavy tile of two": (1, 100, {(0, 70): 20_000}),
    # 4 entries per row on average, no giant row: the fused launch takes the bf16-product kernel (csrc/spmm_gemm.hip
    # spmm_gemm_bx3_kernel) — its heavy tiles first, then the main loop with the next tile drawn under the products, long rows on
    # four of its eight waves
    "heavy tiles among short rows": (2, 20_000, {(0, 70): 20_000, (0, 71): 9_000, (1, 19_999): 12_000, (1, 5): 300, (1, 6): 0}),
}


@pytest.mark.parametrize("case", list(LONG_ROW_CASES))
@pytest.mark.parametrize("F", [64, 128, 256, 520])
def test_spmm_long_rows_split_across_waves(case, F):
    """Rows longer than 256 entries are gathered by the four waves of their block (quarters of the row, partial sums added
    in wave order) and tiles of more than 8192 entries are processed before the counter-driven loop starts
    (csrc/spmm_row.h): against the C oracle's row-by-row sum, and reproducible to the bit."""
    T, N, lengths = LONG_ROW_CASES[case]
    if F >= 256 and N > 10_000:
        pytest.skip("the wide cases are covered on the small shapes")
    csr = _csr_with_row_lengths(T, N, lengths, seed=F + N)
    X = torch.randn(T, N, F, generator=torch.Generator().manual_seed(2))
    A = csr.to(DEV)
    Y1 = ops.kernels.spmm(A, X.to(DEV))
    ref = ref_spmm(csr, X)
    assert_close(Y1, ref, REL_TOL, f"{case} F={F}")
    rows = torch.tensor([k * N + i for (k, i) in lengths])
    assert_close(Y1.reshape(T * N, F)[rows.to(DEV)], ref.reshape(T * N, F)[rows], REL_TOL, f"{case} F={F}: the special rows")
    assert torch.equal(Y1, ops.kernels.spmm(A, X.to(DEV)))


@pytest.mark.parametrize("case", list(LONG_ROW_CASES))
@pytest.mark.parametrize("K,Nf,per_slice", [(128, 128, False), (64, 32, True), (16, 128, False)])
def test_spmm_gemm_long_rows_split_across_waves(case, K, Nf, per_slice):
    """The same rows through the fused kernel (its LDS tile is filled by the split too): SpMM intermediate and product
    against the C oracle, the intermediate bit-equal to the plain kernel's (one row-sum order for both), reproducible."""
    T, N, lengths = LONG_ROW_CASES[case]
    csr = _csr_with_row_lengths(T, N, lengths, seed=K + N)
    g = torch.Generator().manual_seed(K * 3 + Nf)
    X = torch.randn(T, N, K, generator=g)
    W = torch.randn(*((T, K, Nf) if per_slice else (K, Nf)), generator=g) * 0.2
    A = csr.to(DEV)
    Y, AX, _ = ops.kernels.spmm_gemm(A, X.to(DEV), W.to(DEV), want_ax=True)
    ref_ax = ref_spmm(csr, X)
    assert_close(AX, ref_ax, REL_TOL, f"{case}: SpMM intermediate")
    assert_close(Y, ref_gemm(ref_ax, W, False, per_slice), REL_TOL, f"{case}: fused K={K} Nf={Nf}")
    assert torch.equal(AX, ops.kernels.spmm(A, X.to(DEV))), "fused and plain kernels sum a row in different orders"
    Y2, _, _ = ops.kernels.spmm_gemm(A, X.to(DEV), W.to(DEV))
    assert torch.equal(Y, Y2)


@pytest.mark.parametrize("F", [16, 24, 64, 128, 200, 256])
def test_short_tiles_mix_with_the_other_walks(F):
    """Tiles on both sides of the short-tile threshold (512 entries per 64 rows, csrc/spmm_row.h) in ONE launch: slices whose
    rows hold 0-3 entries (entry-major walk; whole tiles of empty rows; a ragged last tile: N = 333), slices of 9-30 per row
    (row per wave), a tile at exactly 512 and one at 513 entries, a short tile holding one 200-entry row, and a tile with a
    300-entry row (four-wave split) — plain kernel vs the C oracle, the fused kernel's SpMM intermediate bit-equal to it
    (one row-sum order whichever kernel and weight layout), reruns bit-equal."""
    T, N = 4, 333
    g = torch.Generator().manual_seed(F)
    cnt = torch.zeros(T, N, dtype=torch.int64)
    cnt[0] = torch.randint(0, 4, (N,), generator=g)
    cnt[0, 64:128] = 0                                   # a tile of empty rows
    cnt[1] = torch.randint(9, 31, (N,), generator=g)
    cnt[2] = 8
    cnt[2, 0:64] = 8                                     # exactly 512 entries
    cnt[2, 64:128] = 8
    cnt[2, 64] = 9                                       # 513: the other side of the threshold
    cnt[2, 128:192] = 1
    cnt[2, 130] = 200                                    # a short tile with one long-ish row (<= 256)
    cnt[3] = 2
    cnt[3, 200] = 300                                    # > 256: the four-wave split inside an otherwise short tile
    rowptr = torch.zeros(T * N + 1, dtype=torch.int64)
    torch.cumsum(cnt.reshape(-1), 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    csr = BatchedCSR(rowptr, torch.randint(0, N, (nnz,), generator=g, dtype=torch.int32), torch.randn(nnz, generator=g), T, N)
    X = torch.randn(T, N, F, generator=g)
    A = csr.to(DEV)
    Y1 = ops.kernels.spmm(A, X.to(DEV))
    assert_close(Y1, ref_spmm(csr, X), REL_TOL, f"mixed tiles F={F}")
    assert torch.equal(Y1, ops.kernels.spmm(A, X.to(DEV)))
    if F <= 128 and F % 8 == 0:
        for per_slice in (False, True):
            W = torch.randn(*((T, F, 40) if per_slice else (F, 40)), generator=g) * 0.2
            Y, AX, _ = ops.kernels.spmm_gemm(A, X.to(DEV), W.to(DEV), want_ax=True)
            assert torch.equal(AX, Y1), "fused and plain kernels sum a row in different orders"
            assert_close(Y, ref_gemm(ref_spmm(csr, X), W, False, per_slice), REL_TOL, f"mixed tiles fused F={F}")


@pytest.mark.parametrize("case", ["threshold", "one heavy tile of two"])
@pytest.mark.parametrize("F,Nf", [(2, 6), (6, 6), (8, 4), (3, 16)])
def test_narrow_kernels_leave_long_rows_to_the_whole_wave(case, F, Nf):
    """The narrow kernels (F <= 8: G lanes per row) hand rows of more than 32 trips to all 64 lanes of the wave
    (csrc/spmm_row.h: narrow_wave_row): plain SpMM and the fused narrow SpMM + GEMM against the C oracle, reproducible."""
    T, N, lengths = LONG_ROW_CASES[case]
    csr = _csr_with_row_lengths(T, N, lengths, seed=F * 13 + Nf)
    g = torch.Generator().manual_seed(F)
    X = torch.randn(T, N, F, generator=g)
    W = torch.randn(F, Nf, generator=g) * 0.5
    A = csr.to(DEV)
    Y = ops.kernels.spmm(A, X.to(DEV))
    ref = ref_spmm(csr, X)
    assert_close(Y, ref, REL_TOL, f"narrow spmm F={F} {case}")
    assert torch.equal(Y, ops.kernels.spmm(A, X.to(DEV)))
    if ops.kernels.spmm_gemm_supported(F, Nf):
        Z, AX, _ = ops.kernels.spmm_gemm(A, X.to(DEV), W.to(DEV), act="selu", want_ax=True)
        assert_close(AX, ref, REL_TOL, f"narrow fused: SpMM intermediate F={F} {case}")
        assert_close(Z, torch.nn.functional.selu(ref_gemm(ref, W)), REL_TOL, f"narrow fused F={F} Nf={Nf} {case}")
        assert torch.equal(Z, ops.kernels.spmm_gemm(A, X.to(DEV), W.to(DEV), act="selu")[0])


GIANT_ROWS = {(0, 17): 1_000_000, (0, 18): 40_000, (0, 90_000): 32_769, (1, 50_000): 32_768, (1, 50_001): 65_536,
              (1, 100_002): 200_001, (1, 0): 20_000}


@pytest.mark.parametrize("F", [16, 128, 320])
def test_spmm_giant_rows_are_summed_chunk_by_chunk(F):
    """Rows of more than 32 768 entries take the giant-row plan (csr.BatchedCSR.giant_plan -> tmgcn_spmm_csr_batched_f32_plan):
    4 096-entry chunks summed by a launch in front of the main kernel, partial sums added in chunk order.  Against the C oracle,
    bit-reproducible, and within fp32 summation order of the plan-less route (four waves of one block per row)."""
    T, N = 2, 100_003
    csr = _csr_with_row_lengths(T, N, GIANT_ROWS, seed=F)
    A = csr.to(DEV)
    rows, chunks = A.giant_plan()
    want_rows = sorted(k * N + i for (k, i), n in GIANT_ROWS.items() if n > 32_768)
    assert rows.tolist() == want_rows and chunks.numel() == len(want_rows) + 1 + int(chunks[len(want_rows)])
    assert int(chunks[len(want_rows)]) == sum((n + 4_095) // 4_096 for n in GIANT_ROWS.values() if n > 32_768)
    X = torch.randn(T, N, F, generator=torch.Generator().manual_seed(2))
    Xd = X.to(DEV)
    Y = ops.kernels.spmm(A, Xd)
    ref = ref_spmm(csr, X)
    assert_close(Y, ref, REL_TOL, f"giant rows F={F}")
    special = torch.tensor([k * N + i for (k, i) in GIANT_ROWS])
    assert_close(Y.reshape(T * N, F)[special.to(DEV)], ref.reshape(T * N, F)[special], REL_TOL, f"giant rows F={F}: the special rows")
    assert torch.equal(Y, ops.kernels.spmm(A, Xd))
    Y0 = ops.kernels.ops.spmm_csr_batched(A.rowptr, A.col, A.val, Xd, N, A.avg_nnz_per_row, None, None)     # no plan
    assert_close(Y0, ref, REL_TOL, "plan-less route")
    other = torch.ones(T * N, dtype=torch.bool)
    other[torch.tensor(want_rows)] = False
    assert torch.equal(Y.reshape(T * N, F)[other.to(DEV)], Y0.reshape(T * N, F)[other.to(DEV)])          # only the giant rows differ


@pytest.mark.parametrize("K,Nf,per_slice", [(128, 128, False), (64, 32, True)])
def test_spmm_gemm_giant_rows(K, Nf, per_slice):
    """The same plan through the fused kernel (forward operand) and, by autograd, through its transposed operand."""
    T, N = 2, 100_003
    csr = _csr_with_row_lengths(T, N, GIANT_ROWS, seed=K)
    g = torch.Generator().manual_seed(K * 3 + Nf)
    X = torch.randn(T, N, K, generator=g)
    W = torch.randn(*((T, K, Nf) if per_slice else (K, Nf)), generator=g) * 0.2
    A = csr.to(DEV)
    Y, AX, _ = ops.kernels.spmm_gemm(A, X.to(DEV), W.to(DEV), want_ax=True)
    ref_ax = ref_spmm(csr, X)
    assert_close(AX, ref_ax, REL_TOL, "giant rows: SpMM intermediate")
    assert_close(Y, ref_gemm(ref_ax, W, False, per_slice), REL_TOL, f"giant rows: fused K={K} Nf={Nf}")
    assert torch.equal(AX, ops.kernels.spmm(A, X.to(DEV))), "fused and plain kernels sum a giant row in different orders"
    assert torch.equal(Y, ops.kernels.spmm_gemm(A, X.to(DEV), W.to(DEV))[0])
    if not per_slice:        # autograd: the TRANSPOSED operand has its own plan (giant COLUMNS of Â: here none — hubs gather, they are not gathered)
        Xg, Wg = X.to(DEV).requires_grad_(True), W.to(DEV).requires_grad_(True)
        dY = torch.randn(T, N, Nf, generator=g)
        ops.spmm_feature_gemm(A, Xg, Wg).backward(dY.to(DEV))
        assert_close(Xg.grad, ref_spmm(csr.transpose(), ref_gemm(dY, W, trans_w=True)), REL_TOL, "dX")
        # and with the roles swapped — the transpose as the forward operand: its transpose (= csr) brings the giant plan to the backward
        At = csr.transpose().to(DEV)
        assert At.giant_plan()[0] is None and At.transpose().giant_plan()[0] is not None
        Xg2 = X.to(DEV).requires_grad_(True)
        ops.spmm_feature_gemm(At, Xg2, Wg.detach()).backward(dY.to(DEV))
        assert_close(Xg2.grad, ref_spmm(csr, ref_gemm(dY, W, trans_w=True)), REL_TOL, "dX through the giant plan of the transpose")


@pytest.mark.parametrize("symmetric", [False, True])
def test_layer_on_powerlaw_graph_vs_oracle(symmetric):
    """The S4-shaped layer (F 128 -> 128) on synth.device_powerlaw_csr (capped-Zipf row lengths; the cap reaches N here):
    forward and both gradients of ops.spmm_feature_gemm against the C oracle chain, fused and unfused."""
    T, N, F = 2, 20_000, 128
    csr = synth.device_powerlaw_csr(T, N, 32, "cpu", symmetric=symmetric)
    cnt = csr.rowptr[1:] - csr.rowptr[:-1]
    assert int(cnt.max()) > 5_000 and float(cnt.float().median()) < 40
    g = torch.Generator().manual_seed(5)
    X0, W0 = torch.rand(T, N, F, generator=g), torch.randn(F, F, generator=g) * 0.1
    dY = torch.randn(T, N, F, generator=g)
    ax = ref_spmm(csr, X0)
    y_ref = ref_gemm(ax, W0)
    dx_ref = ref_spmm(csr.transpose(), ref_gemm(dY, W0, trans_w=True))
    dw_ref = (ax.double().reshape(-1, F).t() @ dY.double().reshape(-1, F)).float()
    A = csr.to(DEV)
    for fuse in (True, False):
        X, W = X0.to(DEV).requires_grad_(True), W0.to(DEV).requires_grad_(True)
        Y = ops.spmm_feature_gemm(A, X, W, fuse=fuse)
        Y.backward(dY.to(DEV))
        assert_close(Y.detach(), y_ref, REL_TOL, f"Y fuse={fuse}")
        assert_close(X.grad, dx_ref, REL_TOL, f"dX fuse={fuse}")
        assert_close(W.grad, dw_ref, REL_TOL, f"dW fuse={fuse}")


def test_spmm_transpose_is_adjoint():
    T, N, F = 3, 120, 16
    csr = rand_csr(T, N, 6.0, seed=3).to(DEV)
    X = torch.randn(T, N, F, device=DEV)
    Yb = torch.randn(T, N, F, device=DEV)
    lhs = (ops.kernels.spmm(csr, X).double() * Yb.double()).sum()      # fp64 inner products: the identity, not fp32 summation noise
    rhs = (X.double() * ops.kernels.spmm(csr.transpose(), Yb).double()).sum()
    scale = float((ops.kernels.spmm(csr, X).double() * Yb.double()).abs().sum())
    assert abs(float(lhs - rhs)) <= 1e-5 * scale


def test_spmm_empty_matrix():
    csr = BatchedCSR.from_coo(torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long),
                              torch.zeros(0, dtype=torch.long), torch.zeros(0), 2, 10).to(DEV)
    Y = ops.kernels.spmm(csr, torch.randn(2, 10, 16, device=DEV))
    assert float(Y.abs().max()) == 0.0


# ------------------------------------------------------------------------------------- P1
def ref_mt(M64, X, transpose=False):
    lib = load_c_oracle()
    T = X.shape[0]
    Y = torch.empty_like(X)
    lib.ref_mtransform(cptr(M64), T, int(transpose), cptr(X), cptr(Y), X.numel() // T)
    return Y


@pytest.mark.parametrize("T,b", [(1, 1), (5, 3), (34, 20), (95, 20), (128, 20), (150, 7), (40, 40)])
@pytest.mark.parametrize("N,F", [(37, 2), (64, 16), (11, 3)])
@pytest.mark.parametrize("transpose", [False, True])
def test_mtransform_band(T, b, N, F, transpose):
    M = torch.from_numpy(synth.band_M(T, b, "matlab")).contiguous()
    op = ops.MOperator(M, DEV)
    assert op.band_lo == min(b, T) - 1 and op.band_hi == 0
    X = torch.randn(T, N, F, generator=torch.Generator().manual_seed(T + N))
    Y = ops.kernels.mtransform(op, X.to(DEV), transpose=transpose)
    assert_close(Y, ref_mt(M, X, transpose), REL_TOL, f"mtransform T={T} b={b}")


@pytest.mark.parametrize("T", [6, 64, 130])
def test_mtransform_dense_and_inverse_roundtrip(T):
    g = torch.Generator().manual_seed(T)
    M = (torch.randn(T, T, generator=g, dtype=torch.float64) / T ** 0.5 + torch.eye(T, dtype=torch.float64) * 2).contiguous()
    op = ops.MOperator(M, DEV)
    X = torch.randn(T, 33, 4, generator=g)
    Y = ops.kernels.mtransform(op, X.to(DEV))
    assert_close(Y, ref_mt(M, X), REL_TOL, "dense M")
    back = ops.kernels.mtransform(op.inverse(), Y)
    assert_close(back, X, 1e-5, "Minv∘M = I")
    # the reference's band M (read_data.m:116-124) has a dense lower-triangular inverse
    Mb = torch.from_numpy(synth.band_M(T, 20, "matlab"))
    opb = ops.MOperator(Mb, DEV)
    back = ops.kernels.mtransform(opb.inverse(), ops.kernels.mtransform(opb, X.to(DEV)))
    assert_close(back, X, 1e-5, "band Minv∘M = I")


@pytest.mark.parametrize("T,N,F,transpose", [(128, 301, 4, False), (128, 301, 4, True), (100, 77, 8, False),
                                             (37, 1000, 4, True), (128, 16, 4, False), (17, 5, 4, False)])
def test_mtransform_dense_bf16_split_kernel(T, N, F, transpose):
    """Dense operators with T <= 128 run on the bf16 matrix cores after an exact 3-way split of both
    operands: fp32 accuracy against the fp64 oracle on operands spanning six orders of magnitude,
    tail column tiles (C not a multiple of 64), tail rows (T not a multiple of 16 / 32), Mᵀ, and
    bit-reproducible."""
    g = torch.Generator().manual_seed(T * 7 + N)
    M = (torch.randn(T, T, generator=g, dtype=torch.float64) * torch.exp(torch.randn(T, 1, generator=g, dtype=torch.float64) * 2)).contiguous()
    op = ops.MOperator(M, DEV)
    assert op.band_lo + op.band_hi + 1 > 20                      # not the band kernel
    X = (torch.randn(T, N, F, generator=g) * torch.exp(torch.randn(1, N, 1, generator=g) * 3)).contiguous()
    Y = ops.kernels.mtransform(op, X.to(DEV), transpose=transpose)
    ref = ref_mt(M, X, transpose)
    # per-column scale: every column's error against that column's own magnitude
    err = ((Y.cpu().double() - ref.double()).abs().amax(0) / ref.double().abs().amax(0).clamp_min(1e-30)).max()
    assert float(err) <= REL_TOL, float(err)
    assert torch.equal(Y, ops.kernels.mtransform(op, X.to(DEV), transpose=transpose))


def test_mtransform_dense_bf16_split_windows_and_grouped_rows():
    """The dense bf16-split kernel with operator windows (a rank's own output slices of the gathered
    tensor, and the adjoint) and with the group-interleaved row storage of the all-to-all layouts."""
    T, N, F, tl = 96, 50, 4, 8
    g = torch.Generator().manual_seed(5)
    M = (torch.randn(T, T, generator=g, dtype=torch.float64) / T ** 0.5 + torch.eye(T, dtype=torch.float64)).contiguous()
    op = ops.MOperator(M, DEV)
    X = torch.randn(T, N, F, generator=g)
    full = ref_mt(M, X)
    part = ops.kernels.mtransform(op, X.to(DEV), row_off=24, col_off=0, T_out=24)
    assert_close(part, full[24:48], REL_TOL, "dense row window")
    dY = torch.randn(24, N, F, generator=g)
    pad = torch.zeros(T, N, F)
    pad[24:48] = dY
    dX = ops.kernels.mtransform(op, dY.to(DEV), transpose=True, row_off=0, col_off=24, T_out=T)
    assert_close(dX, ref_mt(M, pad, True), REL_TOL, "dense adjoint window")
    # group-interleaved storage: logical row k lives at (k % tl) * (T / tl) + k // tl
    pos = torch.tensor([(k % tl) * (T // tl) + k // tl for k in range(T)])
    Yg = ops.kernels.mtransform(op, X.to(DEV), y_group_rows=tl)
    assert_close(Yg.cpu()[pos], full, REL_TOL, "dense, grouped output rows")
    Xg = torch.empty_like(X)
    Xg[pos] = X
    Yx = ops.kernels.mtransform(op, Xg.to(DEV), x_group_rows=tl)
    assert_close(Yx, full, REL_TOL, "dense, grouped input rows")


def test_mtransform_windowed_rows():
    """row/col offsets: a rank computing only its own slices from the full X, and the adjoint."""
    T, N, F = 24, 40, 4
    M = torch.from_numpy(synth.band_M(T, 6, "matlab")).contiguous()
    op = ops.MOperator(M, DEV)
    X = torch.randn(T, N, F, generator=torch.Generator().manual_seed(9))
    full = ref_mt(M, X)
    part = ops.kernels.mtransform(op, X.to(DEV), row_off=8, col_off=0, T_out=8)
    assert_close(part, full[8:16], REL_TOL, "row window")
    # adjoint of that window: dX[j] = Σ_{k in window} M[8+k][j] dY[k]
    dY = torch.randn(8, N, F, generator=torch.Generator().manual_seed(10))
    pad = torch.zeros(T, N, F)
    pad[8:16] = dY
    dX = ops.kernels.mtransform(op, dY.to(DEV), transpose=True, row_off=0, col_off=8, T_out=T)
    assert_close(dX, ref_mt(M, pad, True), REL_TOL, "adjoint window")


@pytest.mark.parametrize("T,band,F", [(24, 6, 4), (24, 6, 3),      # band kernel: float4 and scalar (C % 4 != 0) forms
                                      (96, None, 4),                # dense bf16-split kernel
                                      (160, None, 4)])              # exact-f32 matrix-core kernel (T_in > 128)
def test_mtransform_column_window_is_bitwise_the_contiguous_product(T, band, F):
    """tmgcn_mtransform_ld_f32 (ops.kernels.mtransform_out): a chunk of columns transformed into /
    out of a column window of a wider tensor — the consumer of the node-chunked all-gather.  Per
    output element it must be the SAME arithmetic as the contiguous product: bit-equal, for the
    forward row window, for the adjoint, and with the group-interleaved row storage on either side."""
    N, Tl, k0, tl = 52, 8, 8, 8
    g = torch.Generator().manual_seed(11)
    if band:
        M = torch.from_numpy(synth.band_M(T, band, "matlab")).contiguous()
    else:
        M = (torch.randn(T, T, generator=g, dtype=torch.float64) / T ** 0.5 + torch.eye(T, dtype=torch.float64)).contiguous()
    op = ops.MOperator(M, DEV)
    X = torch.randn(T, N, F, generator=g).to(DEV)
    K = ops.kernels
    whole = K.mtransform(op, X, row_off=k0, col_off=0, T_out=Tl)                 # [Tl, N, F]
    dY = torch.randn(Tl, N, F, generator=g).to(DEV)
    whole_T = K.mtransform(op, dY, transpose=True, row_off=0, col_off=k0, T_out=T)  # [T, N, F]
    pos = torch.tensor([(k % tl) * (T // tl) + k // tl for k in range(T)], device=DEV)
    Xg = torch.empty_like(X)
    Xg[pos] = X                                                                   # grouped storage of the input rows
    for chunks in ([(0, N)], [(0, 20), (20, 40), (40, N)], [(0, 1), (1, N)]):
        out = torch.full((Tl, N, F), float("nan"), device=DEV)
        out_T = torch.full((T, N, F), float("nan"), device=DEV)
        for c0, c1 in chunks:
            # forward: contiguous (gathered) chunk -> window of the resident result
            K.mtransform_out(op, Xg[:, c0:c1].contiguous(), out[:, c0:c1], row_off=k0, col_off=0, x_group_rows=tl)
            # adjoint: window of the upstream gradient -> contiguous send buffer, grouped rows
            buf = torch.empty(T, c1 - c0, F, device=DEV)
            K.mtransform_out(op, dY[:, c0:c1], buf, transpose=True, row_off=0, col_off=k0, y_group_rows=tl)
            out_T[:, c0:c1] = buf[pos]
        assert torch.equal(out, whole), f"forward window differs, chunks={chunks}"
        assert torch.equal(out_T, whole_T), f"adjoint window differs, chunks={chunks}"
    with pytest.raises(RuntimeError):                                              # slices must be dense [n, F] blocks
        K.mtransform_out(op, X.transpose(1, 2), torch.empty(Tl, F, N, device=DEV), row_off=k0)


# ------------------------------------------------------------------------------------- P3
def ref_gemm(A, W, trans_w=False, per_slice=False):
    lib = load_c_oracle()
    T, N, K = A.shape
    Nf = W.shape[-2] if trans_w else W.shape[-1]
    Y = torch.empty(T, N, Nf)
    lib.ref_gemm(cptr(A), cptr(W), cptr(Y), T * N, K, Nf, int(trans_w), N if per_slice else 0,
                 W.shape[-1] * W.shape[-2] if per_slice else 0)
    return Y


@pytest.mark.parametrize("K,Nf", [(2, 6), (6, 6), (6, 2), (12, 2), (16, 16), (128, 128), (128, 64), (100, 50),
                                  (32, 200), (300, 40), (7, 33), (64, 5),
                                  (132, 64), (256, 128), (384, 100), (512, 36), (516, 16)])   # k-chunked split kernel; 516: past it
@pytest.mark.parametrize("per_slice", [False, True])
@pytest.mark.parametrize("trans_w", [False, True])
def test_gemm(K, Nf, per_slice, trans_w):
    T, N = 3, 150
    g = torch.Generator().manual_seed(K * 31 + Nf)
    A = torch.randn(T, N, K, generator=g)
    wshape = ((Nf, K) if trans_w else (K, Nf))
    W = torch.randn(*((T,) + wshape if per_slice else wshape), generator=g)
    Y = ops.kernels.gemm(A.to(DEV), W.to(DEV), trans_w=trans_w)
    assert_close(Y, ref_gemm(A, W, trans_w, per_slice), REL_TOL, f"gemm {K}x{Nf}")


@pytest.mark.parametrize("K,Nf", [(2, 6), (6, 2), (16, 16), (128, 128), (128, 64), (64, 100), (100, 50), (36, 200),
                                  (300, 40), (256, 128), (132, 32)])
@pytest.mark.parametrize("per_slice", [False, True])
@pytest.mark.parametrize("trans_w", [False, True])
def test_gemm_bf16_stored_weight(K, Nf, per_slice, trans_w):
    """tmgcn_gemm_bf16w_f32 (the "bf16 weights" configuration): W enters in bf16, A / Y / accumulation stay
    fp32.  Against the C oracle on the widened weight at the fp32 tolerance, and bit-identical to the fp32
    entry on the widened weight in every kernel (the split kernel only drops products with zero planes)."""
    T, N = 3, 150
    g = torch.Generator().manual_seed(K * 17 + Nf)
    A = torch.randn(T, N, K, generator=g)
    wshape = ((Nf, K) if trans_w else (K, Nf))
    W = torch.randn(*((T,) + wshape if per_slice else wshape), generator=g).to(torch.bfloat16)
    Y = ops.kernels.gemm(A.to(DEV), W.to(DEV), trans_w=trans_w)
    assert Y.dtype == torch.float32
    assert_close(Y, ref_gemm(A, W.float(), trans_w, per_slice), REL_TOL, f"bf16-W gemm {K}x{Nf}")
    assert torch.equal(Y, ops.kernels.gemm(A.to(DEV), W.float().to(DEV), trans_w=trans_w)), "differs from the widened weight"


def test_gemm_bf16_stored_weight_autograd_and_ragged_tiles():
    """Gradients with a bf16-stored parameter: dA through the bf16-W kernel (Wᵀ), dW summed in fp32 and
    rounded once to the parameter's dtype; fused activation; a row count that leaves a ragged last tile."""
    g = torch.Generator().manual_seed(5)
    A = torch.randn(2, 4099, 128, generator=g).to(DEV)
    W = (torch.randn(128, 96, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    dY = torch.randn(2, 4099, 96, generator=g).to(DEV)
    out = {}
    for name, w in (("bf16", W), ("wide", W.float())):
        a, w = A.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = ops.feature_gemm(a, w, act="leaky")
        y.backward(dY)
        out[name] = (y.detach(), a.grad, w.grad)
    assert out["bf16"][2].dtype == torch.bfloat16
    assert torch.equal(out["bf16"][0], out["wide"][0]) and torch.equal(out["bf16"][1], out["wide"][1])
    assert torch.equal(out["bf16"][2], out["wide"][2].to(torch.bfloat16))
    ref = torch.nn.functional.leaky_relu(A.double() @ W.double(), 0.01)
    assert max_rel_err(out["bf16"][0], ref.float()) <= REL_TOL


@pytest.mark.parametrize("act", ["relu", "leaky", "selu"])
@pytest.mark.parametrize("K,Nf", [(2, 6), (128, 128), (256, 64), (200, 130)])
def test_gemm_fused_activation(act, K, Nf):
    from oracle import tmgcn_oracle as orc
    A = torch.randn(2, 100, K, generator=torch.Generator().manual_seed(1))
    W = torch.randn(K, Nf, generator=torch.Generator().manual_seed(2))
    Y, pre = ops.kernels.gemm(A.to(DEV), W.to(DEV), act=act, want_pre=True)
    ref_pre = ref_gemm(A, W)
    assert_close(pre, ref_pre, REL_TOL, "pre-activation")
    assert_close(Y, orc.ACTS[act](pre.cpu()), 2e-6, "activation of the kernel's own pre-activation")
    dy = torch.randn_like(ref_pre)
    x = pre.cpu().clone().requires_grad_(True)
    orc.ACTS[act](x).backward(dy)
    assert_close(ops.kernels.act_bwd(pre, dy.to(DEV), act), x.grad, 2e-6, "activation backward")
    assert_close(ops.kernels.act_fwd(pre, act), orc.ACTS[act](pre.cpu()), 2e-6, "activation forward")


@pytest.mark.parametrize("K,Nf", [(2, 6), (6, 6), (12, 2), (16, 16), (128, 128), (100, 50), (300, 40), (64, 5)])
@pytest.mark.parametrize("per_slice", [False, True])
def test_gemm_dw(K, Nf, per_slice):
    lib = load_c_oracle()
    T, N = 3, 333
    g = torch.Generator().manual_seed(K + Nf)
    A = torch.randn(T, N, K, generator=g)
    dY = torch.randn(T, N, Nf, generator=g)
    ref = torch.empty((T, K, Nf) if per_slice else (K, Nf))
    lib.ref_gemm_dw(cptr(A), cptr(dY), cptr(ref), T * N, K, Nf, N if per_slice else 0)
    got = ops.kernels.gemm_dw(A.to(DEV), dY.to(DEV), per_slice)
    assert_close(got, ref, REL_TOL, f"dW {K}x{Nf}")
    assert torch.equal(got, ops.kernels.gemm_dw(A.to(DEV), dY.to(DEV), per_slice)), "dW not reproducible"


@pytest.mark.parametrize("K,Nf", [(2, 6), (6, 6), (6, 2), (2, 2), (8, 8), (4, 6)])
@pytest.mark.parametrize("T,N", [(3, 333), (1, 7), (5, 4099), (95, 6000)])
def test_gemm_dw_narrow_shapes_and_sizes(K, Nf, T, N):
    """The narrow dW kernel (four lanes per row, four rows in flight, reduced by its own last block) over row counts
    from less than one block trip to hundreds of slabs, shared and per-slice weights, against the C oracle."""
    lib = load_c_oracle()
    g = torch.Generator().manual_seed(K * 100 + Nf + N)
    A = torch.randn(T, N, K, generator=g)
    dY = torch.randn(T, N, Nf, generator=g)
    for per_slice in (False, True):
        ref = torch.empty((T, K, Nf) if per_slice else (K, Nf))
        lib.ref_gemm_dw(cptr(A), cptr(dY), cptr(ref), T * N, K, Nf, N if per_slice else 0)
        got = ops.kernels.gemm_dw(A.to(DEV), dY.to(DEV), per_slice)
        assert_close(got, ref, REL_TOL, f"dW {K}x{Nf} T={T} N={N} per_slice={per_slice}")
        assert torch.equal(got, ops.kernels.gemm_dw(A.to(DEV), dY.to(DEV), per_slice)), "dW not reproducible"


@pytest.mark.parametrize("act", ["relu", "leaky", "selu"])
@pytest.mark.parametrize("K,Nf,per_slice", [(2, 6, False), (6, 6, False), (2, 6, True)])
def test_gemm_dw_with_folded_activation_gradient(act, K, Nf, per_slice):
    """dW = Aᵀ·(dY ⊙ act'(pre)) in one launch == act_bwd followed by the plain dW kernel (bit for bit: the same
    products in the same order), and the layer-1 backward of feature_gemm(act=...) uses it when the input is a constant."""
    from tmgcn_amd import _lib
    T, N = 4, 2500
    g = torch.Generator().manual_seed(K + Nf)
    A, dY = torch.randn(T, N, K, generator=g).to(DEV), torch.randn(T, N, Nf, generator=g).to(DEV)
    pre = torch.randn(T, N, Nf, generator=g).to(DEV)
    fused = ops.kernels.ops.bgemm_dW_act(A, dY, pre, _lib.ACT_IDS[act], per_slice)
    two = ops.kernels.gemm_dw(A, ops.kernels.act_bwd(pre, dY, act), per_slice)
    assert torch.equal(fused, two)
    # through autograd: constant input, differentiable weight
    W = (torch.randn(*((T,) if per_slice else ()), K, Nf, generator=g) * 0.5).to(DEV).requires_grad_(True)
    Y = ops.feature_gemm(A, W, act=act)
    Y.backward(dY)
    W2 = W.detach().clone().requires_grad_(True)
    A2 = A.clone().requires_grad_(True)                       # input needs a gradient too: the unfused route
    ops.feature_gemm(A2, W2, act=act).backward(dY)
    assert torch.equal(W.grad, W2.grad)


@pytest.mark.parametrize("K,Nf,R", [(128, 128, 70001), (132, 260, 5000), (16, 16, 33), (64, 36, 4099), (20, 100, 777)])
def test_gemm_dw_bf16_split_is_fp32_accurate(K, Nf, R):
    """dW runs on the bf16 matrix cores through an exact 3-way split of the fp32 operands (hi + mid +
    lo planes, six plane products per term).  Against an fp64 product, on operands whose rows span
    six orders of magnitude, it must be as accurate as the exact-f32 MFMA kernel it replaces — both
    are selectable per call (algo = TMGCN_DW_F32MFMA | TMGCN_DW_AUTO) — and bit-reproducible."""
    g = torch.Generator().manual_seed(R)
    A = (torch.randn(1, R, K, generator=g) * torch.exp(torch.randn(1, R, 1, generator=g) * 3)).to(DEV)
    dY = torch.randn(1, R, Nf, generator=g).to(DEV)
    ref = A[0].double().T @ dY[0].double()
    err = {}
    for mode, algo in ((0, "f32mfma"), (1, "auto")):
        got = ops.kernels.gemm_dw(A, dY, False, algo=algo)
        err[mode] = max_rel_err(got, ref)
        assert torch.equal(got, ops.kernels.gemm_dw(A, dY, False, algo=algo)), "dW not reproducible"
    assert err[1] <= 2e-6 and err[1] <= 2 * err[0] + 1e-7, err


def test_grid_reserve_is_per_launch_not_process_wide():
    """Two layers with different reserves in one process: each launch gets its own grid (read back
    from the launch through the kernel timer hook), results identical, and an unsharded launch that
    follows a reserved one is back at the full grid — there is no process-wide setting (ABI v2)."""
    from tmgcn_amd import _lib
    lib = _lib.load()
    assert not hasattr(lib, "tmgcn_config_set")
    A = rand_csr(2, 40000, 9.0, 3).to(DEV)
    g = torch.Generator().manual_seed(5)
    X = torch.randn(2, 40000, 64, generator=g).to(DEV)
    W = torch.randn(64, 64, generator=g).to(DEV)
    y0, _, _ = ops.kernels.spmm_gemm(A, X, W)
    y1, _, _ = ops.kernels.spmm_gemm(A, X, W, grid_reserve=256)
    y2, _, _ = ops.kernels.spmm_gemm(A, X, W, grid_reserve=0)
    assert torch.equal(y0, y1) and torch.equal(y0, y2)
    with pytest.raises(RuntimeError):
        ops.kernels.spmm_gemm(A, X, W, grid_reserve=-1)


def test_gemm_many_rows_persistent_loop():
    """More tiles than the persistent grid (1024 blocks x 64 rows)."""
    T, N, K, Nf = 2, 70000, 32, 32
    g = torch.Generator().manual_seed(4)
    A = torch.randn(T, N, K, generator=g)
    W = torch.randn(K, Nf, generator=g)
    assert_close(ops.kernels.gemm(A.to(DEV), W.to(DEV)), ref_gemm(A, W), REL_TOL, "persistent gemm")


# ------------------------------------------------------------------------------------- errors
def test_errors_are_runtime_errors():
    with pytest.raises(RuntimeError):
        ops.kernels.spmm(rand_csr(2, 10, 2.0, 1).to(DEV), torch.randn(2, 11, 4, device=DEV))
    with pytest.raises(RuntimeError):
        ops.kernels.spmm(rand_csr(2, 10, 2.0, 1).to(DEV), torch.randn(2, 10, 4))  # CPU tensor
    with pytest.raises(RuntimeError):
        ops.kernels.gemm(torch.randn(2, 10, 4, device=DEV), torch.randn(5, 3, device=DEV))
    with pytest.raises(RuntimeError):
        ops.kernels.spmm(rand_csr(2, 10, 2.0, 1).to(DEV), torch.randn(2, 10, 4, device=DEV).double())


# ------------------------------------------------------------------------------------- P2+P3 fused
@pytest.mark.parametrize("K", [16, 24, 32, 48, 64, 96, 120, 128])
@pytest.mark.parametrize("Nf", [1, 6, 32, 100, 128])
@pytest.mark.parametrize("trans_w,per_slice", [(False, False), (True, False), (False, True)])
def test_spmm_gemm_fused(K, Nf, trans_w, per_slice):
    T, N = 3, 130  # N not a multiple of the 64-row tile: per-slice tiles are clipped
    csr = rand_csr(T, N, 9.0, seed=K + Nf)
    g = torch.Generator().manual_seed(K * 3 + Nf)
    X = torch.randn(T, N, K, generator=g)
    wshape = (Nf, K) if trans_w else (K, Nf)
    W = torch.randn(*((T,) + wshape if per_slice else wshape), generator=g)
    assert ops.kernels.spmm_gemm_supported(K, Nf)
    Y, AX, _ = ops.kernels.spmm_gemm(csr.to(DEV), X.to(DEV), W.to(DEV), trans_w=trans_w, want_ax=True)
    ref_ax = ref_spmm(csr, X)
    assert_close(AX, ref_ax, REL_TOL, "fused: SpMM intermediate")
    assert_close(Y, ref_gemm(ref_ax, W, trans_w, per_slice), REL_TOL, f"fused K={K} Nf={Nf}")
    Y2, _, _ = ops.kernels.spmm_gemm(csr.to(DEV), X.to(DEV), W.to(DEV), trans_w=trans_w)
    assert torch.equal(Y, Y2), "fused kernel not reproducible"


@pytest.mark.parametrize("act", [None, "relu", "leaky", "selu"])
@pytest.mark.parametrize("want_pre", [False, True])
@pytest.mark.parametrize("N,K,Nf,deg", [(130, 128, 128, 3.0), (77, 64, 40, 20.0), (64, 32, 7, 6.0),
                                        (130, 128, 126, 3.0), (77, 64, 40, 3.0), (200, 128, 30, 2.0), (64, 64, 128, 5.0)])
def test_spmm_gemm_wide_epilogue_variants(act, want_pre, N, K, Nf, deg):
    """The wide fused kernel's epilogue (round 6: scalar bases, the activation decoded once — none at all for act = None —,
    the row guard only in the half tile a slice ends in; csrc/spmm_gemm.hip fused_store_half): every activation, with and
    without the pre-activation output, slices that end inside a half tile (N = 130, 77) and on its edge (64), output widths
    that do and do not fill the waves' 32-column strips — Y, pre and AX against the oracle, reruns bit-equal.  The cases of
    fewer than 14 entries per row at K = 128 / 64 run the bf16-product kernel (spmm_gemm_bx3_kernel): its Y tile leaves
    through LDS as whole rows when a row is a whole number of float4 (Nf = 128, 40), from the accumulators otherwise (126, 30)."""
    from oracle import tmgcn_oracle as orc
    T = 3
    csr = rand_csr(T, N, deg, seed=N + K)
    g = torch.Generator().manual_seed(N * 3 + Nf)
    X = torch.randn(T, N, K, generator=g)
    W = torch.randn(K, Nf, generator=g) * 0.3
    Y, AX, pre = ops.kernels.spmm_gemm(csr.to(DEV), X.to(DEV), W.to(DEV), act=act, want_ax=True, want_pre=want_pre)
    ref_ax = ref_spmm(csr, X)
    ref_pre = ref_gemm(ref_ax, W, False, False)
    assert_close(AX, ref_ax, REL_TOL, "wide fused: SpMM intermediate")
    if want_pre and act:                                   # (without an activation Y IS the pre-activation: the op returns none)
        assert_close(pre, ref_pre, REL_TOL, "wide fused: pre-activation")
    else:
        assert pre is None
    want = orc.ACTS[act](ref_pre) if act else ref_pre
    assert_close(Y, want, REL_TOL, f"wide fused: act={act}")
    Y2, _, pre2 = ops.kernels.spmm_gemm(csr.to(DEV), X.to(DEV), W.to(DEV), act=act, want_pre=want_pre)
    assert torch.equal(Y, Y2) and (pre is None or torch.equal(pre, pre2))


@pytest.mark.parametrize("act", [None, "selu"])
def test_spmm_gemm_autograd_matches_unfused(act):
    """Fused op (backward = (ÂᵀdY)Wᵀ) against the two-kernel path (backward = Âᵀ(dY Wᵀ))."""
    T, N, K, Nf = 4, 200, 32, 64
    csr = rand_csr(T, N, 8.0, seed=11).to(DEV)
    g = torch.Generator().manual_seed(12)
    X0 = torch.randn(T, N, K, generator=g).to(DEV)
    W0 = (torch.randn(K, Nf, generator=g) * 0.2).to(DEV)
    dY = torch.randn(T, N, Nf, generator=g).to(DEV)
    res = []
    for fuse in (True, False):
        X = X0.clone().requires_grad_(True)
        W = W0.clone().requires_grad_(True)
        Y = ops.spmm_feature_gemm(csr, X, W, act=act, fuse=fuse)
        Y.backward(dY)
        res.append((Y.detach(), X.grad, W.grad))
    for a, b, what in zip(res[0], res[1], ("Y", "dX", "dW")):
        assert_close(a, b, REL_TOL, f"fused vs unfused {what} act={act}")


@pytest.mark.parametrize("K,Nf", [(1, 1), (2, 6), (6, 6), (6, 2), (3, 16), (8, 5), (4, 7)])
@pytest.mark.parametrize("trans_w,per_slice", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("act", [None, "selu"])
def test_spmm_gemm_fused_small_widths(K, Nf, trans_w, per_slice, act):
    """The reference's real widths (2 -> 6 -> 6): one launch for P2+P3(+P5)."""
    from oracle import tmgcn_oracle as orc
    T, N = 3, 130
    csr = rand_csr(T, N, 9.0, seed=K * 17 + Nf)
    g = torch.Generator().manual_seed(K + Nf)
    X = torch.randn(T, N, K, generator=g)
    wshape = (Nf, K) if trans_w else (K, Nf)
    W = torch.randn(*((T,) + wshape if per_slice else wshape), generator=g)
    assert ops.kernels.spmm_gemm_supported(K, Nf)
    Y, AX, pre = ops.kernels.spmm_gemm(csr.to(DEV), X.to(DEV), W.to(DEV), trans_w=trans_w, act=act, want_ax=True, want_pre=True)
    ref_ax = ref_spmm(csr, X)
    ref_pre = ref_gemm(ref_ax, W, trans_w, per_slice)
    assert_close(AX, ref_ax, REL_TOL, "small fused: SpMM intermediate")
    if act:
        assert_close(pre, ref_pre, REL_TOL, "small fused: pre-activation")
        assert_close(Y, orc.ACTS[act](pre.cpu()), 2e-6, "small fused: activation")
    else:
        assert_close(Y, ref_pre, REL_TOL, f"small fused K={K} Nf={Nf}")


def test_spmm_gemm_unsupported_width_raises():
    csr = rand_csr(2, 20, 3.0, seed=1).to(DEV)
    assert not ops.kernels.spmm_gemm_supported(20, 8)
    with pytest.raises(RuntimeError):
        ops.kernels.spmm_gemm(csr, torch.randn(2, 20, 20, device=DEV), torch.randn(20, 8, device=DEV))
    # the dispatcher falls back to the two-kernel path
    Y = ops.spmm_feature_gemm(csr, torch.randn(2, 20, 20, device=DEV), torch.randn(20, 8, device=DEV))
    assert tuple(Y.shape) == (2, 20, 8)


# ------------------------------------------------------------------------------------- P4 edge head
@pytest.mark.parametrize("F,C", [(2, 2), (6, 2), (6, 3), (16, 8), (32, 2), (5, 1), (128, 2), (100, 3), (256, 8),
                                 (4, 4), (8, 1), (64, 2), (192, 2), (20, 3)])   # + every narrow / wide kernel instantiation family
@pytest.mark.parametrize("E", [0, 1, 1000, 70001])
def test_edge_head_fwd_bwd(F, C, E):
    T, N = 3, 97
    g = torch.Generator().manual_seed(F * 100 + C + E)
    Z = torch.randn(T, N, F, generator=g)
    U = torch.randn(2 * F, C, generator=g)
    edges = torch.stack([torch.randint(0, T, (E,), generator=g), torch.randint(0, N, (E,), generator=g),
                         torch.randint(0, N, (E,), generator=g)])
    if E > 10:
        edges[:, 1] = edges[:, 0]  # a duplicated edge
        edges[1, 2:9] = 5          # a hub row
    dout = torch.randn(E, C, generator=g)
    # fp64 reference of the reference's statements (ehf:228-232) and autograd through them
    Zr = Z.double().clone().requires_grad_(True)
    Ur = U.double().clone().requires_grad_(True)
    Zf = Zr.reshape(-1, F)
    ref = torch.cat((Zf[edges[0] * N + edges[1]], Zf[edges[0] * N + edges[2]]), dim=1) @ Ur
    ref.backward(dout.double())
    eidx = ops.EdgeIndex(edges, N, DEV)
    Zg = Z.to(DEV).requires_grad_(True)
    Ug = U.to(DEV).requires_grad_(True)
    assert ops.kernels.edge_head_supported(F, C)
    out = ops.edge_head(Zg, eidx, Ug, fuse=True)
    out.backward(dout.to(DEV))
    if E:
        assert_close(out, ref.detach(), REL_TOL, "edge head logits")
        assert_close(Zg.grad, Zr.grad, REL_TOL, "edge head dZ")
        assert_close(Ug.grad, Ur.grad, REL_TOL, "edge head dU")
    else:
        assert out.shape == (0, C) and float(Zg.grad.abs().max()) == 0.0 and float(Ug.grad.abs().max()) == 0.0
    # bitwise reproducible (no atomics), and identical to the unfused torch path within tolerance
    Z2 = Z.to(DEV).requires_grad_(True)
    U2 = U.to(DEV).requires_grad_(True)
    ops.edge_head(Z2, eidx, U2, fuse=True).backward(dout.to(DEV))
    assert torch.equal(Z2.grad, Zg.grad) and torch.equal(U2.grad, Ug.grad)


def test_edge_index_is_validated_before_any_launch():
    """The forward kernel gathers Z[t*N+node] unchecked, so the index is checked once per edge set
    (ADVICE r1): slice >= T, node >= N or a negative entry raise (IndexError, as the reference's
    ``Y.reshape(-1,F)[idx]`` does, and RuntimeError for callers that catch that)."""
    ok = torch.tensor([[0, 1], [1, 2], [3, 4]])
    ops.EdgeIndex(ok, 10, DEV, T=2)
    for bad in ([[0, 2], [1, 2], [3, 4]], [[0, 1], [1, 10], [3, 4]], [[0, 1], [1, 2], [3, -1]], [[-1, 1], [1, 2], [3, 4]]):
        for dev in ("cpu", DEV):
            with pytest.raises(IndexError):
                ops.EdgeIndex(torch.tensor(bad).to(dev), 10, DEV, T=2)
            with pytest.raises(RuntimeError):
                ops.EdgeIndex(torch.tensor(bad).to(dev), 10, DEV, T=2)
    with pytest.raises(RuntimeError):
        ops.EdgeIndex(torch.zeros(2, 5, dtype=torch.int64), 10, DEV, T=2)


def test_edge_head_wide_falls_back():
    assert not ops.kernels.edge_head_supported(300, 2)
    Z = torch.randn(2, 10, 300, device=DEV)
    U = torch.randn(600, 2, device=DEV)
    e = ops.EdgeIndex(torch.tensor([[0, 1], [1, 2], [3, 4]]), 10, DEV)
    assert tuple(ops.edge_head(Z, e, U).shape) == (2, 2)
    with pytest.raises(RuntimeError):
        ops.edge_head(Z, e, U, fuse=True)


# ------------------------------------------------------------------------------------- weighted CE (opt-in)
def test_weighted_cross_entropy_ignore_index_and_corrupt_labels():
    """ignore_index targets are skipped exactly as torch skips them; any OTHER label outside [0, C)
    (torch device-asserts there) makes the loss and every gradient NaN — loud, never a silently
    smaller training set."""
    from tmgcn_amd.losses import WeightedCrossEntropy
    g = torch.Generator().manual_seed(3)
    E, C = 5000, 3
    z = torch.randn(E, C, generator=g)
    t = torch.randint(0, C, (E,), generator=g)
    w = torch.rand(C, generator=g) + 0.1
    t_ign = t.clone()
    t_ign[::5] = -100
    zr = z.double().clone().requires_grad_(True)
    ref = torch.nn.CrossEntropyLoss(weight=w.double())(zr, t_ign)
    ref.backward()
    zg = z.to(DEV).requires_grad_(True)
    loss = WeightedCrossEntropy(w)(zg, t_ign.to(DEV))
    loss.backward()
    assert abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref))
    assert_close(zg.grad, zr.grad, 1e-6, "wce grad with ignore_index")
    assert float(zg.grad[::5].abs().max()) == 0.0
    for corrupt in (C, -1, 7):
        t_bad = t.clone()
        t_bad[17] = corrupt
        zb = z.to(DEV).requires_grad_(True)
        lb = WeightedCrossEntropy(w)(zb, t_bad.to(DEV))
        lb.backward()
        assert torch.isnan(lb) and torch.isnan(zb.grad).all(), corrupt
    with pytest.raises(RuntimeError):                                   # an in-range class cannot be the ignored one
        WeightedCrossEntropy(w, ignore_index=1)(z.to(DEV), t.to(DEV))



@pytest.mark.parametrize("E,C", [(1, 2), (1000, 2), (5000, 3), (300, 8), (3_000_001, 2), (777, 4), (40_001, 4)])
def test_weighted_cross_entropy_matches_torch_fp64(E, C):
    from tmgcn_amd.losses import WeightedCrossEntropy
    g = torch.Generator().manual_seed(E + C)
    z = torch.randn(E, C, generator=g) * 5
    t = torch.randint(0, C, (E,), generator=g)
    w = torch.rand(C, generator=g) + 0.1
    zr = z.double().clone().requires_grad_(True)
    ref = torch.nn.CrossEntropyLoss(weight=w.double())(zr, t)
    (ref * 1.7).backward()
    zg = z.to(DEV).requires_grad_(True)
    loss = WeightedCrossEntropy(w)(zg, t.to(DEV))
    (loss * 1.7).backward()
    assert abs(float(loss) - float(ref)) <= 2e-7 * max(1.0, abs(float(ref)))
    assert_close(zg.grad, zr.grad, 2e-6, "wce dlogits")
    with pytest.raises(RuntimeError):
        WeightedCrossEntropy(torch.ones(9))(torch.randn(4, 9, device=DEV), torch.zeros(4, dtype=torch.long, device=DEV))



def _oracle_layer12(H, W1, act1, A, W2, act2, dZ, dtype=torch.float32):
    """The oracle's restatement of what ops.layer12 computes — layers 1 + 2 of the narrow models the reference's way
    (ehf:330-335 then the default branch ehf:348-349; EmbeddingKWGCN ehf:486-487): fp32 H·W1, the non-linearity,
    `.double()`, one fp64 sparse.mm per slice into an fp32 buffer (oracle.slice_spmm), fp32 ·W2, autograd for dW1 / dW2.
    dtype=float64: the same math with fp64 weights and buffers (the "truth" fp32 reduction noise is judged against)."""
    from oracle import tmgcn_oracle as orc
    At = A.to_coo_list(torch.float64)
    w1, w2 = W1.detach().cpu().to(dtype).requires_grad_(True), W2.detach().cpu().to(dtype).requires_grad_(True)
    orc.BUFFER_DTYPE = dtype
    try:
        Y = torch.matmul(H.detach().cpu().to(dtype), w1)
        Y = (orc.ACTS[act1](Y) if act1 else Y).double()
        Z = torch.matmul(orc.slice_spmm(At, Y), w2)
        Z = orc.ACTS[act2](Z) if act2 else Z
        Z.backward(dZ.detach().cpu().to(dtype))
    finally:
        orc.BUFFER_DTYPE = torch.float32
    return Z.detach(), w1.grad, w2.grad


def _assert_layer12_vs_oracle(got, H, W1, act1, A, W2, act2, dZ, tag):
    """The bar of tests/test_gpu_configs.py: within 1e-5 of the reference-way fp32 oracle — or, only where that fp32
    result is ITSELF more than 1e-5 from the fp64 truth of the same math (dW1 / dW2 are sums over T·N rows: the
    reference's fp32 reduction order is good to a few 1e-5 at 30 000 rows), within 1e-6 of the truth and at least ten
    times closer to it than the reference is."""
    ref32 = _oracle_layer12(H, W1, act1, A, W2, act2, dZ)
    truth = None
    for x, y, what, k in zip(got, ref32, ("Z", "dW1", "dW2"), range(3)):
        if float(y.abs().max()) == 0.0:
            assert float(x.abs().max()) == 0.0, f"{tag} {what}: the oracle's result is all zero"
            continue
        e_ref = max_rel_err(x, y)
        if e_ref <= REL_TOL:
            continue
        truth = truth or _oracle_layer12(H, W1, act1, A, W2, act2, dZ, torch.float64)
        e_truth, e_ref_truth = max_rel_err(x, truth[k]), max_rel_err(y, truth[k])
        assert e_ref_truth > REL_TOL and e_truth <= 1e-6 and 10 * e_truth <= e_ref_truth, \
            f"{tag} {what}: vs the oracle (fp32) {e_ref:.2e}, vs the fp64 truth {e_truth:.2e} (the oracle itself {e_ref_truth:.2e})"



@pytest.mark.parametrize("T,N,nnz", [(1, 300, 0), (3, 256, 0), (1, 300, 5), (2, 256, 700), (3, 257, 900), (1, 256, 768),
                                     (5, 1024, 15360), (2, 255, 600), (1, 4096, 20000), (9, 513, 4617)])
def test_layer12_edge_shapes(T, N, nnz):
    """Shapes at the seams of the row walks of csrc/layer12.hip: empty adjacencies, slices of exactly / just above / just
    below 256 nodes (the entry-major kernels' block of rows), a single slice, a ragged last block."""
    from tmgcn_amd import adjacency
    rng = np.random.default_rng(T * 7919 + N)
    A = adjacency.DeviceCOO.from_edges(rng.integers(0, T, nnz), rng.integers(0, N, nnz), rng.integers(0, N, nnz),
                                       rng.uniform(0.1, 1.0, nnz).astype(np.float32), T, N).sort_reduce().to_csr()
    g = torch.Generator().manual_seed(7)
    H = torch.randn(T, N, 2, generator=g).to(DEV)
    W1, W2 = (torch.randn(2, 6, generator=g) * 0.7).to(DEV), (torch.randn(6, 6, generator=g) * 0.7).to(DEV)
    dZ = torch.randn(T, N, 6, generator=g).to(DEV)
    outs = []
    for fuse in (True, False):
        a1, a2 = W1.clone().requires_grad_(True), W2.clone().requires_grad_(True)
        Z = ops.layer12(H, a1, "selu", A, a2, None, fuse=fuse)
        Z.backward(dZ)
        outs.append((Z.detach(), a1.grad, a2.grad))
    for x, y, what in zip(outs[0], outs[1], ("Z", "dW1", "dW2")):      # the bit-level check: HIP fused vs HIP two-operator
        if float(y.abs().max()) == 0.0:
            assert float(x.abs().max()) == 0.0, what
        else:
            assert_close(x, y, 2e-6, what)
    for o, tag in zip(outs, ("fused", "two-operator")):                # the parity check: each against the oracle
        _assert_layer12_vs_oracle(o, H, W1, "selu", A, W2, None, dZ, tag)


@pytest.mark.parametrize("act2", [None, "selu"])
def test_layer12_entry_major_kernels_on_skewed_rows(act2):
    """The entry-major layer kernels (sparse rows: a block walks the contiguous entry range of its 256 rows, tile by
    tile) where a block's rows hold several tiles of entries and others almost none — 250 hub rows of ~13 entries at
    the start of every slice of 1 300 nodes, one or two entries elsewhere, a slice boundary inside a block, a ragged
    last block — against the two-operator route; reproducible."""
    from tmgcn_amd import adjacency
    rng = np.random.default_rng(11)
    T, N = 3, 1300
    ks, is_, js = [], [], []
    for t in range(T):
        hub_i = np.repeat(np.arange(250), 13)
        ks += [np.full(hub_i.size, t), np.full(N, t)]
        is_ += [hub_i, np.arange(N)]
        js += [rng.integers(0, N, hub_i.size), np.arange(N)]
    k, i, j = np.concatenate(ks), np.concatenate(is_), np.concatenate(js)
    A = adjacency.DeviceCOO.from_edges(k, i, j, rng.uniform(0.1, 1.0, k.size).astype(np.float32), T, N).sort_reduce().to_csr()
    assert A.avg_nnz_per_row < 4 and int((A.rowptr[256] - A.rowptr[0])) > 2 * 1024      # one lane per row regime; three tiles
    g = torch.Generator().manual_seed(9)
    H = torch.randn(T, N, 2, generator=g).to(DEV)
    W1, W2 = (torch.randn(2, 6, generator=g) * 0.7).to(DEV), (torch.randn(6, 6, generator=g) * 0.7).to(DEV)
    dZ = torch.randn(T, N, 6, generator=g).to(DEV)
    out = []
    for fuse in (True, False, True):
        a1, a2 = W1.clone().requires_grad_(True), W2.clone().requires_grad_(True)
        Z = ops.layer12(H, a1, "selu", A, a2, act2, fuse=fuse)
        Z.backward(dZ)
        out.append((Z.detach(), a1.grad, a2.grad))
    assert_close(out[0][0], out[1][0], 2e-6, "Z")
    assert_close(out[0][1], out[1][1], 2e-6, "dW1")
    assert_close(out[0][2], out[1][2], 2e-6, "dW2")
    assert all(torch.equal(x, y) for x, y in zip(out[0], out[2]))
    _assert_layer12_vs_oracle(out[0], H, W1, "selu", A, W2, act2, dZ, "entry-major fused")


@pytest.mark.parametrize("base_deg", [1, 12])            # the backward's two walks: entry-major (sparse rows) / lanes per row
@pytest.mark.parametrize("act2", [None, "selu"])
def test_layer12_hub_rows(base_deg, act2):
    """Hub rows (3 000 and 700 entries next to rows of 1-12) through the fused layers 1 + 2: the entry-major kernels hand a long
    segment of a tile to the owning wave, the lanes-per-row backward hands a long row to the whole wave (csrc/layer12.hip) —
    forward and both weight gradients against the oracle, reproducible."""
    T, N = 3, 1300
    lengths = {(0, 5): 3000, (1, 700): 3000, (1, 701): 700, (2, 1299): 1500, (2, 0): 65}
    g0 = torch.Generator().manual_seed(base_deg)
    cnt = torch.full((T * N,), base_deg, dtype=torch.int64)
    for (k, i), n in lengths.items():
        cnt[k * N + i] = n
    rowptr = torch.zeros(T * N + 1, dtype=torch.int64)
    torch.cumsum(cnt, 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    rows = torch.repeat_interleave(torch.arange(T * N), cnt)
    cols = torch.randint(0, N, (nnz,), generator=g0)
    from tmgcn_amd import adjacency
    A = adjacency.DeviceCOO.from_edges((rows // N).numpy(), (rows % N).numpy(), cols.numpy(),
                                       (torch.rand(nnz, generator=g0) * 0.5 + 0.1).numpy().astype(np.float32), T, N).sort_reduce().to_csr()
    assert int((A.rowptr[1:] - A.rowptr[:-1]).max()) > 1000
    g = torch.Generator().manual_seed(7)
    H = torch.randn(T, N, 2, generator=g).to(DEV)
    W1, W2 = (torch.randn(2, 6, generator=g) * 0.7).to(DEV), (torch.randn(6, 6, generator=g) * 0.7).to(DEV)
    dZ = torch.randn(T, N, 6, generator=g).to(DEV)
    outs = []
    for _ in range(2):
        a1, a2 = W1.clone().requires_grad_(True), W2.clone().requires_grad_(True)
        Z = ops.layer12(H, a1, "selu", A, a2, act2, fuse=True)
        Z.backward(dZ)
        outs.append((Z.detach(), a1.grad, a2.grad))
    assert all(torch.equal(x, y) for x, y in zip(*outs))
    _assert_layer12_vs_oracle(outs[0], H, W1, "selu", A, W2, act2, dZ, f"hub rows, {base_deg} per row")


@pytest.mark.parametrize("T,N,deg,skew", [(5, 250_000, 2.0, False),       # 4 883 row blocks of 256 rows: more slabs than thread blocks stay resident
                                          (3, 30_000, 10.0, True),         # hub rows: a partition of (first row, rows) pairs, heaviest first
                                          (2, 1500, 2.0, False)])          # 12 row blocks: fewer than the sixteen hand-off groups
def test_layer12_backward_draws_row_blocks_reproducibly(T, N, deg, skew):
    """The entry-major backward (csrc/layer12.hip): resident thread blocks DRAW the row blocks they work on, so which block
    sums which rows differs from run to run — dW1 must not: it is summed per row block and the row blocks' sums in a fixed
    order.  Ten runs bit-equal, and equal to the unfused route (GEMM + fused SpMM + their autograd) to fp32 rounding."""
    if skew:
        A = synth.device_powerlaw_csr(T, N, int(deg), DEV, first_slice=3, hub_cap=4000, symmetric=True)
        assert A.transpose().is_skewed() and A.transpose().row_blocks() is not None
    else:
        A = synth.device_er_csr(T, N, int(deg), DEV, first_slice=3)
    g = torch.Generator().manual_seed(11)
    H = torch.randn(T, N, 2, generator=g).to(DEV)
    W1, W2 = (torch.randn(2, 6, generator=g) * 0.7).to(DEV), (torch.randn(6, 6, generator=g) * 0.7).to(DEV)
    dZ = torch.randn(T, N, 6, generator=g).to(DEV)
    dZ[:, N // 3:] = 0                                    # (as in training: most rows of the embedding's gradient are zero)
    runs = []
    for _ in range(10):
        a1, a2 = W1.clone().requires_grad_(True), W2.clone().requires_grad_(True)
        ops.layer12(H, a1, "selu", A, a2, None, fuse=True).backward(dZ)
        runs.append((a1.grad.clone(), a2.grad.clone()))
    assert all(torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1]) for r in runs[1:])
    b1, b2 = W1.clone().requires_grad_(True), W2.clone().requires_grad_(True)
    ops.layer12(H, b1, "selu", A, b2, None, fuse=False).backward(dZ)
    for got, ref, name in ((runs[0][0], b1.grad, "dW1"), (runs[0][1], b2.grad, "dW2")):
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err <= 2e-6, (name, err)


@pytest.mark.parametrize("act1,act2", [("selu", None), ("relu", None), ("leaky", "relu"), (None, "selu")])
@pytest.mark.parametrize("T,N,deg,F,Nf", [(5, 300, 3.0, 6, 6), (3, 77, 12.0, 6, 2), (4, 500, 0.4, 2, 6), (2, 64, 40.0, 8, 4),
                                          # staged variants (slice in LDS): several blocks per slice with a ragged last
                                          # chunk, many slices, and a slice just too large for them (72 KB)
                                          (7, 1000, 27.0, 6, 6), (150, 200, 9.0, 6, 6), (2, 3000, 9.0, 6, 6),
                                          # at the 64 KB edge of the staged variants: the slice alone is exactly 64 KB / just
                                          # under it, but slice + row-pointer slab + static arrays are not (unstaged kernels then)
                                          (2, 2048, 9.0, 8, 8), (2, 2700, 9.0, 6, 6), (3, 2300, 9.0, 6, 6)])
def test_layer12_fused_matches_the_two_operators(act1, act2, T, N, deg, F, Nf):
    """ops.layer12 (csrc/layer12.hip: layers 1 + 2 of the narrow 2-layer models in one launch each way) against
    feature_gemm followed by spmm_feature_gemm: the same Z to the last bit or two (same per-lane fmaf chains; a row's
    partial sums are folded over however many lanes each kernel gives a row), dW1 / dW2 to fp32 rounding (the fused
    backward accumulates dW1 per row in fp64), reproducible."""
    from tmgcn_amd import adjacency
    rng = np.random.default_rng(T * 1000 + N)
    nnz = int(T * N * deg)
    A = adjacency.DeviceCOO.from_edges(rng.integers(0, T, nnz), rng.integers(0, N, nnz), rng.integers(0, N, nnz),
                                       rng.uniform(0.1, 1.0, nnz).astype(np.float32), T, N).sort_reduce().to_csr()
    g = torch.Generator().manual_seed(7)
    H = torch.randn(T, N, 2, generator=g).to(DEV)
    W1 = (torch.randn(2, F, generator=g) * 0.7).to(DEV)
    W2 = (torch.randn(F, Nf, generator=g) * 0.7).to(DEV)
    dZ = torch.randn(T, N, Nf, generator=g).to(DEV)
    a1, a2 = W1.clone().requires_grad_(True), W2.clone().requires_grad_(True)
    Z = ops.layer12(H, a1, act1, A, a2, act2, fuse=True)
    Z.backward(dZ)
    b1, b2 = W1.clone().requires_grad_(True), W2.clone().requires_grad_(True)
    Zr = ops.layer12(H, b1, act1, A, b2, act2, fuse=False)
    Zr.backward(dZ)
    # two fp32 summation orders of a row (27 terms at the densest case: each lane of the fused kernels adds CONSECUTIVE
    # non-zeros, the two-operator route strided ones): 1.0e-6 measured there, 3e-7 on the 3-per-row cases
    assert_close(Z.detach(), Zr.detach(), 2e-6, "Z")
    assert_close(a1.grad, b1.grad, 2e-6, "dW1")
    assert_close(a2.grad, b2.grad, 2e-6, "dW2")
    _assert_layer12_vs_oracle((Z.detach(), a1.grad, a2.grad), H, W1, act1, A, W2, act2, dZ, "fused")
    c1, c2 = W1.clone().requires_grad_(True), W2.clone().requires_grad_(True)
    ops.layer12(H, c1, act1, A, c2, act2, fuse=True).backward(dZ)
    assert torch.equal(c1.grad, a1.grad) and torch.equal(c2.grad, a2.grad)
    with torch.no_grad():                                      # no gradients: nothing but Z is stored
        assert torch.equal(ops.layer12(H, W1, act1, A, W2, act2), Z.detach())
