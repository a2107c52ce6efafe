"""CPU: the seeded synthetic S4 adjacencies (tmgcn_amd/synth.py) — the balanced one of SURVEY §8d and the skewed one the
roofline_skewed bench leg runs on."""
import numpy as np
import torch

from tmgcn_amd import synth


def _rows(A):
    return (A.rowptr[1:] - A.rowptr[:-1]).reshape(A.T, A.N)


def test_powerlaw_degrees_shape():
    d = synth.powerlaw_degrees(2_000_000, 32)
    assert d[0] == 100_000 and 5 <= (d == 100_000).sum() <= 30            # a handful of hubs at the cap
    assert abs(d.sum() / 2e6 - 32) < 1.0 and d[-1] >= 1 and np.all(np.diff(d) <= 0)
    share = np.cumsum(d) / d.sum()
    assert 0.55 < share[200_000] < 0.70                                   # a tenth of the rows holds ~60 % of the entries
    assert synth.powerlaw_degrees(500, 32)[0] == 500                      # cap = N on small graphs


def test_powerlaw_csr_is_seeded_per_slice_sorted_and_row_normalised():
    T, N = 3, 4000
    for sym in (False, True):
        A = synth.device_powerlaw_csr(T, N, 32, "cpu", first_slice=5, symmetric=sym)
        cnt = _rows(A)
        assert int(cnt.min()) >= 1 and int(cnt.max()) > 1000 and abs(float(cnt.float().mean()) - 33) < 2.5
        rid = A.row_ids()
        col = A.col.long()
        # columns ascending inside every row; the self loop is there; values = 1 / row length
        same = rid[1:] == rid[:-1]
        assert bool((col[1:][same] >= col[:-1][same]).all())
        assert bool(torch.zeros(T * N, dtype=torch.bool).index_put_((rid[col == rid % N],), torch.tensor(True)).all())
        assert torch.allclose(torch.zeros(T * N, dtype=torch.float64).index_add_(0, rid, A.val.double()), torch.ones(T * N, dtype=torch.float64), atol=1e-5)
        # slice k depends on first_slice + k only: any rank can regenerate any slice
        one = synth.device_powerlaw_csr(1, N, 32, "cpu", first_slice=6, symmetric=sym)
        a, b = int(A.rowptr[N]), int(A.rowptr[2 * N])
        assert torch.equal(one.col, A.col[a:b]) and torch.equal(one.val, A.val[a:b])
        assert torch.equal(one.rowptr, A.rowptr[N:2 * N + 1] - a)
        if sym:                                                         # pattern symmetric: the transpose has the same row lengths
            assert torch.equal(_rows(A.transpose()), cnt)
        else:                                                           # skewed out-degree only: in-degrees are Poisson-like
            assert int(_rows(A.transpose()).max()) < 120
    assert synth.device_csr("er", 1, 50, 4, "cpu").nnz == 50 * 5


def test_row_blocks_partition_by_entries():
    """csr.BatchedCSR.row_blocks (the partition the entry-major layer kernels take): (first row, rows) pairs that tile the rows, heaviest block first, at most 256 rows
    and at most max_entries + the longest row of entries per block, cut at row boundaries; None when nothing needs cutting."""
    from tmgcn_amd.csr import BatchedCSR
    assert synth.device_er_csr(2, 1000, 2, "cpu").row_blocks() is None          # 3 per row: 768 entries per 256 rows
    g = torch.Generator().manual_seed(0)
    T, N = 3, 1300
    cnt = torch.randint(0, 4, (T * N,), generator=g)
    cnt[:250] = 13                                                            # dense rows at the start of slice 0
    cnt[N + 100] = 5000                                                       # one row longer than max_entries
    cnt[2 * N - 1] = 0
    rowptr = torch.zeros(T * N + 1, dtype=torch.int64)
    torch.cumsum(cnt, 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    A = BatchedCSR(rowptr, torch.randint(0, N, (nnz,), generator=g, dtype=torch.int32), torch.rand(nnz, generator=g), T, N)
    pairs = A.row_blocks()
    assert pairs is A.row_blocks() and pairs.dtype == torch.int64 and pairs.shape[1] == 2
    ent_listed = rowptr[pairs[:, 0] + pairs[:, 1]] - rowptr[pairs[:, 0]]
    assert bool((ent_listed[1:] <= ent_listed[:-1]).all())                    # the heaviest blocks first
    order = torch.argsort(pairs[:, 0])
    blk = torch.cat((pairs[order, 0], torch.tensor([T * N])))
    assert bool((pairs[order, 0] + pairs[order, 1] == blk[1:]).all())         # the blocks tile the rows: each row exactly once
    assert int(blk[0]) == 0 and int(blk[-1]) == T * N and bool((blk[1:] > blk[:-1]).all())
    rows = blk[1:] - blk[:-1]
    ent = rowptr[blk[1:]] - rowptr[blk[:-1]]
    assert int(rows.max()) <= 256 and int(ent.max()) <= 1024 + int(cnt.max())
    assert int((ent > 1024).sum()) == 1                                       # only the block of the 5 000-entry row
    assert blk.numel() - 1 > (T * N + 255) // 256                             # the dense region was cut further
    # every multiple of 256 rows is still a boundary (blocks are only ever cut, never merged)
    assert set(range(0, T * N, 256)) <= set(blk.tolist())
    tri = A.trivial_row_blocks()
    assert tri.shape == ((T * N + 255) // 256, 2) and int(tri[:, 1].sum()) == T * N and int(tri[:, 1].max()) == 256


def test_tile_block_diagonal_is_a_kronecker_replication():
    """synth.tile_block_diagonal (bench.py --graph chess_tiled: the reference's real operand at bench size): slice s of the
    result is I_reps ⊗ A[slices[s]] — row lengths, values and the order inside a row are kept; the default takes every slice."""
    from tmgcn_amd.csr import BatchedCSR
    g = synth.dynamic_graph(T=4, N=9, edges_per_slice=12, seed=0, no_diag=2)
    A = BatchedCSR.from_coo_list(g.At_list())
    B = synth.tile_block_diagonal(A, 3, [3, 1])
    assert (B.T, B.N, B.nnz) == (2, 27, 3 * int(A.rowptr[4 * 9] - A.rowptr[3 * 9] + A.rowptr[2 * 9] - A.rowptr[9]))
    D, E = A.to_dense(), B.to_dense()
    for s, k in enumerate([3, 1]):
        assert torch.equal(E[s], torch.kron(torch.eye(3, dtype=D.dtype), D[k]))
    cnt = (A.rowptr[1:] - A.rowptr[:-1]).view(4, 9)
    assert torch.equal((B.rowptr[1:] - B.rowptr[:-1]).view(2, 3, 9)[0], cnt[3].expand(3, -1))
    C = synth.tile_block_diagonal(A, 2)
    assert (C.T, C.N, C.nnz) == (4, 18, 2 * A.nnz)
    # the slice choice of the bench leg: global slice g of the tiled tensor is chess slice (5 g + 4) mod 80
    assert [(5 * s + 4) % 80 for s in range(16)] == list(range(4, 80, 5))
