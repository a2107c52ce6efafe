"""CPU: the C-ABI library loads and exports every symbol include/tmgcn.h declares (no compute
calls without a GPU), the ctypes table mirrors the header, and the host-side logic (batched CSR
ingest/transpose/sharding, band detection, the no-CPU-fallback rule) behaves."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from _util import HERE, ROOT, coo_list, golden
import tmgcn_amd
from tmgcn_amd import _lib, ops, synth
from tmgcn_amd.csr import BatchedCSR


def header_functions():
    src = open(os.path.join(ROOT, "include", "tmgcn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tmgcn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 12
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tmgcn.h but not exported by libtmgcn_hip.so"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == header_functions()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "tmgcn.h")).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), f"{name}: header has {len(params)} parameters, ctypes table {len(args)}"


def test_abi_version_and_error_string():
    lib = _lib.load()
    assert lib.tmgcn_abi_version() == 5
    assert isinstance(lib.tmgcn_last_error(), bytes)
    # argument validation happens before any device work: callable without a GPU
    rc = lib.tmgcn_spmm_csr_batched_f32(None, None, None, None, None, 10, 3, 4, None)
    assert rc == -1 and b"multiple" in lib.tmgcn_last_error() or b"null" in lib.tmgcn_last_error()
    assert lib.tmgcn_gemm_dw_workspace_bytes(1000, 6, 6, 0) > 0
    assert lib.tmgcn_spmm_gemm_supported(128, 128) == 1 and lib.tmgcn_spmm_gemm_supported(6, 6) == 1
    assert lib.tmgcn_spmm_gemm_supported(20, 8) == 0 and lib.tmgcn_spmm_gemm_supported(6, 17) == 0


def test_no_cpu_fallback():
    """The product path fails loudly on CPU tensors / without a device."""
    d = golden("g2_gcn_condensed1")
    X = torch.from_numpy(d["X"])
    At = coo_list(d, "At", X.shape[0], X.shape[1])
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU"):
            tmgcn_amd.EmbeddingGCN(At, X, torch.from_numpy(d["edges"]), torch.from_numpy(d["M"]), hidden_feat=[6, 2],
                                   condensed_W=True, use_Minv=False)
    csr = BatchedCSR.from_coo_list(At, N=X.shape[1])
    with pytest.raises(RuntimeError, match="ROCm"):
        ops.kernels.spmm(csr, X.float())
    with pytest.raises(RuntimeError, match="ROCm"):
        ops.kernels.mtransform(ops.MOperator(d["M"], "cpu"), X.float())


def test_batched_csr_ingest_matches_reference_format():
    d = golden("g3_gcn2_default_relu_condensed1")
    X = torch.from_numpy(d["X"])
    T, N = X.shape[0], X.shape[1]
    At = coo_list(d, "At", T, N)
    csr = BatchedCSR.from_coo_list(At, N=N)
    assert csr.T == T and csr.N == N and csr.rowptr.numel() == T * N + 1
    dense = torch.stack([a.to_dense() for a in At]).float()
    assert torch.equal(csr.to_dense(), dense)
    # columns sorted inside every row; rowptr monotone
    assert bool((csr.rowptr[1:] >= csr.rowptr[:-1]).all())
    rid = csr.row_ids()
    same_row = rid[1:] == rid[:-1]
    assert bool((csr.col[1:][same_row] >= csr.col[:-1][same_row]).all())
    # transpose, slices, round trip to the list-of-COO form
    assert torch.equal(csr.transpose().to_dense(), dense.transpose(1, 2))
    assert csr.transpose().transpose() is csr
    assert torch.equal(csr.slices(2, 5).to_dense(), dense[2:5])
    back = csr.to_coo_list()
    assert all(torch.equal(b.to_dense().float(), a.to_dense().float()) for a, b in zip(At, back))


def test_csr_duplicates_empty_rows_and_bad_indices():
    k = torch.tensor([0, 0, 0, 1])
    i = torch.tensor([2, 2, 0, 3])
    j = torch.tensor([1, 1, 3, 3])  # a duplicate entry (2,1) in slice 0
    v = torch.tensor([1.0, 2.0, 4.0, 8.0])
    csr = BatchedCSR.from_coo(k, i, j, v, T=2, N=4)
    assert csr.nnz == 4  # duplicates kept; they sum in the product like uncoalesced COO in sparse.mm
    assert float(csr.to_dense()[0, 2, 1]) == 3.0
    assert int(csr.rowptr[2]) - int(csr.rowptr[1]) == 0  # empty row
    with pytest.raises(RuntimeError, match="out of range"):
        BatchedCSR.from_coo(k, i, torch.tensor([1, 1, 3, 4]), v, T=2, N=4)
    with pytest.raises(RuntimeError):
        BatchedCSR.from_coo_list([], N=4)


def test_moperator_band_detection():
    for T, b in ((34, 20), (5, 3), (10, 1), (8, 30)):
        for kind in ("matlab", "python"):
            op = ops.MOperator(synth.band_M(T, b, kind), "cpu")
            assert (op.band_lo, op.band_hi) == (min(b, T) - 1, 0)
    dense = ops.MOperator(torch.randn(6, 6, dtype=torch.float64), "cpu")
    assert (dense.band_lo, dense.band_hi) == (5, 5)
    inv = ops.MOperator(synth.band_M(12, 4, "matlab"), "cpu").inverse()
    assert inv.band_hi == 0 and inv.band_lo == 11  # lower-triangular, dense below the diagonal
    assert np.allclose(inv.M64.numpy() @ synth.band_M(12, 4, "matlab"), np.eye(12), atol=1e-12)
    w = ops.MOperator(synth.band_M(12, 4, "matlab"), "cpu").window(0, 11)  # the scripts' M[:-1,:-1]
    assert w.T == 11 and w.band_lo == 3


def test_synth_configs_have_the_scripts_shapes():
    g = synth.sbm_dynamic_graph()  # S0: the SBM plumbing config (T=10, N=500, F=16)
    assert g.X.shape == (10, 500, 16) and len(g.Ct) == 10 and g.M.shape == (10, 10)
    assert all((a != a.T).nnz == 0 and a.diagonal().sum() == 0 for a in g.A_raw)  # symmetric, no self loops
    dens_in = g.A_raw[0][:250, :250].nnz / 250 ** 2
    dens_out = g.A_raw[0][:250, 250:].nnz / 250 ** 2
    assert 0.07 < dens_in < 0.13 and 0.005 < dens_out < 0.02  # p_in = .1, p_out = .01
    s1 = synth.dynamic_graph(T=6, N=80, edges_per_slice=40, seed=3)
    # every normalised Â slice has the full diagonal (the reference's size inference needs it)
    assert all(c.diagonal().min() > 0 for c in s1.Ct)
    lp = synth.dynamic_graph(T=6, N=80, edges_per_slice=40, seed=1, neg_per_pos=3)
    assert set(np.unique(lp.labels)) == {0, 1} and lp.edges.shape[0] == 3
    a = synth.device_er_csr(2, 1000, 32, "cpu", first_slice=5)
    b = synth.device_er_csr(1, 1000, 32, "cpu", first_slice=6)
    assert a.nnz == 2 * 1000 * 33 and torch.equal(a.slices(1, 2).col, b.col)  # seeded by global slice index


# ------------------------------------------------------------------ property-based (hypothesis)
from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=80, deadline=None)
@given(st.lists(st.integers(0, 70), min_size=1, max_size=40), st.integers(1, 24), st.integers(0, 5))
def test_head_loss_plan_splits_rows_into_parts_that_tile_them(lengths, split, shift):
    """ops.HeadLossPlan.split_rows (the row list of the one-pass head + loss kernel, include/tmgcn.h `arow`): every active
    row's entry range is tiled by its parts in order, a part holds at most `split` entries, whole rows carry part id 0, the
    parts of split rows are numbered 1, 2, … in list order, and `srow` names each split row with its first part and count."""
    cnt = torch.tensor(lengths, dtype=torch.int64)
    eptr = torch.zeros(len(lengths) + 1, dtype=torch.int64)
    torch.cumsum(cnt, 0, out=eptr[1:])
    active = torch.nonzero(cnt > 0).flatten()
    beg, end = eptr[:-1][active] + shift, eptr[1:][active] + shift
    n_part = torch.clamp((end - beg + split - 1) // split, min=1)
    arow, srow, n_parts = ops.HeadLossPlan.split_rows(active, beg, end, n_part, split)
    assert arow.dtype == torch.int32 and arow.shape == (int(n_part.sum()) if active.numel() else 0, 4)
    a = arow.long()
    assert bool((a[:, 2] - a[:, 1] <= split).all()) and bool((a[:, 2] > a[:, 1]).all())
    pos = 0
    next_id = 1
    for i, r in enumerate(active.tolist()):
        parts = a[pos:pos + int(n_part[i])]
        pos += int(n_part[i])
        assert bool((parts[:, 0] == r).all())
        assert int(parts[0, 1]) == int(beg[i]) and int(parts[-1, 2]) == int(end[i]) and bool((parts[1:, 1] == parts[:-1, 2]).all())
        if len(parts) == 1:
            assert int(parts[0, 3]) == 0
        else:
            assert parts[:, 3].tolist() == list(range(next_id, next_id + len(parts)))
            row = srow.long()[(srow[:, 0] == r).nonzero().flatten()]
            assert row.shape[0] == 1 and row[0].tolist() == [r, next_id - 1, len(parts), 0]
            next_id += len(parts)
    assert n_parts == next_id - 1 and (srow is None) == (n_parts == 0)
    if srow is not None:
        assert srow.shape[0] == int((n_part > 1).sum())


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(0, 40), min_size=1, max_size=700), st.integers(1, 3), st.sampled_from([16, 64, 256]), st.integers(20, 400))
def test_row_block_partition_tiles_the_rows_heaviest_first(lengths, T, max_rows, max_entries):
    """csr.BatchedCSR.row_blocks for any row lengths: (first row, rows) pairs that cover every row exactly once, at most
    max_rows rows and at most max_entries + the longest row of entries each, listed by entries descending; None exactly when
    no block of max_rows consecutive rows exceeds max_entries."""
    N = (len(lengths) + T - 1) // T
    cnt = torch.tensor((lengths + [0] * (T * N))[:T * N], dtype=torch.int64)
    rowptr = torch.zeros(T * N + 1, dtype=torch.int64)
    torch.cumsum(cnt, 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    A = BatchedCSR(rowptr, torch.zeros(nnz, dtype=torch.int32), torch.ones(nnz), T, N)
    pairs = A.row_blocks(max_rows, max_entries)
    whole = [int(rowptr[min(s + max_rows, T * N)] - rowptr[s]) for s in range(0, T * N, max_rows)]
    assert (pairs is None) == (max(whole) <= max_entries)
    for p in ([] if pairs is None else [pairs]) + [A.trivial_row_blocks(max_rows)]:
        first, rows = p[:, 0], p[:, 1]
        ent = rowptr[first + rows] - rowptr[first]
        assert bool((rows >= 1).all()) and int(rows.max()) <= max_rows
        assert bool((ent[1:] <= ent[:-1]).all())       # heaviest first (the forward's starting order)
        covered = torch.zeros(T * N, dtype=torch.int64)
        for f, n in p.tolist():
            covered[f:f + n] += 1
        assert bool((covered == 1).all())
    if pairs is not None:
        ent = rowptr[pairs[:, 0] + pairs[:, 1]] - rowptr[pairs[:, 0]]
        assert int(ent.max()) <= max_entries + int(cnt.max())
    # the backward's form: the same row blocks as min(16, n) runs of equal length — neighbouring rows, heaviest first inside a run
    lst = A.row_block_runs(max_rows, max_entries)
    want = pairs if pairs is not None else A.trivial_row_blocks(max_rows)
    assert sorted(map(tuple, lst.tolist())) == sorted(map(tuple, want.tolist()))
    ent = rowptr[lst[:, 0] + lst[:, 1]] - rowptr[lst[:, 0]]
    n, runs, last = lst.shape[0], min(16, lst.shape[0]), -1
    for g in range(runs):
        e, f = ent[n * g // runs:n * (g + 1) // runs], lst[n * g // runs:n * (g + 1) // runs, 0]
        assert bool((e[1:] <= e[:-1]).all())
        if f.numel():
            assert int(f.min()) > last
            last = int(f.max())


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 4), st.integers(1, 9), st.lists(st.tuples(st.integers(0, 3), st.integers(0, 8), st.integers(0, 8),
                                                                  st.floats(-4, 4, allow_nan=False, width=32)), max_size=60))
def test_batched_csr_properties(T, N, entries):
    """For any batched COO (duplicates, empty rows/slices, any order): the CSR reproduces the dense
    scatter-add, rowptr is monotone with rowptr[-1] = nnz, columns are sorted inside rows, the
    transpose is the dense transpose, slices() and to_coo_list() round-trip."""
    entries = [(k % T, i % N, j % N, v) for k, i, j, v in entries]
    k = torch.tensor([e[0] for e in entries], dtype=torch.long)
    i = torch.tensor([e[1] for e in entries], dtype=torch.long)
    j = torch.tensor([e[2] for e in entries], dtype=torch.long)
    v = torch.tensor([e[3] for e in entries], dtype=torch.float32)
    csr = BatchedCSR.from_coo(k, i, j, v, T, N)
    dense = torch.zeros(T, N, N)
    dense.index_put_((k, i, j), v, accumulate=True)
    assert torch.allclose(csr.to_dense(), dense, atol=1e-5)
    assert int(csr.rowptr[0]) == 0 and int(csr.rowptr[-1]) == csr.nnz == len(entries)
    assert bool((csr.rowptr[1:] >= csr.rowptr[:-1]).all())
    rid = csr.row_ids()
    same = rid[1:] == rid[:-1]
    assert bool((csr.col[1:][same] >= csr.col[:-1][same]).all())
    assert torch.allclose(csr.transpose().to_dense(), dense.transpose(1, 2), atol=1e-5)
    for a in range(T):
        assert torch.allclose(csr.slices(a, a + 1).to_dense()[0], dense[a], atol=1e-5)
    back = csr.to_coo_list(torch.float32)
    assert all(torch.allclose(b.to_dense(), dense[t], atol=1e-5) for t, b in enumerate(back))
    views = csr.slice_views()
    assert sum(vw.nnz for vw in views) == csr.nnz and all(vw.T == 1 for vw in views)


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 40), st.integers(0, 45), st.integers(0, 45))
def test_moperator_band_is_tight(T, lo, hi):
    """band_lo / band_hi are exactly the extent of the non-zeros, for any banded operator."""
    M = torch.zeros(T, T, dtype=torch.float64)
    for a in range(T):
        for b in range(max(0, a - lo), min(T, a + hi + 1)):
            M[a, b] = 1.0 + a + b
    op = ops.MOperator(M, "cpu")
    assert op.band_lo == min(lo, T - 1) and op.band_hi == min(hi, T - 1)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: include/tmgcn.h compiles as C99 and a C program links against the
    library and calls it (argument validation only — no device work without a GPU)."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "tmgcn.h"\n#include <stdio.h>\n'
                   'int main(void) {\n'
                   '  if (tmgcn_abi_version() != 5) return 1;\n'
                   '  if (tmgcn_spmm_gemm_supported(128, 128) != 1) return 2;\n'
                   '  if (tmgcn_gemm_f32(0, 0, 0, 0, 10, 0, 4, 0, 0, 0, 0, 0, 0) != TMGCN_ERR_INVALID) return 3;\n'
                   '  printf("%s\\n", tmgcn_last_error());\n  return 0;\n}\n')
    exe = tmp_path / "abi"
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L", lib_dir, "-ltmgcn_hip", f"-Wl,-rpath,{lib_dir}"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gemm" in out.stdout


def test_edge_index_validation_on_the_host():
    """EdgeIndex checks (slice, src, dst) against T and N where the edges live — no device needed."""
    import pytest
    import torch
    from tmgcn_amd import ops
    ok = torch.tensor([[0, 1, 1], [1, 2, 0], [3, 4, 9]])
    e = ops.EdgeIndex(ok, 10, "cpu", T=2)
    assert e.src.tolist() == [1, 12, 10] and e.dst.tolist() == [3, 14, 19]
    for bad in ([[2], [0], [0]], [[0], [10], [0]], [[0], [0], [-1]], [[-1], [0], [0]]):
        with pytest.raises(IndexError):
            ops.EdgeIndex(torch.tensor(bad), 10, "cpu", T=2)
    ops.EdgeIndex(torch.zeros(3, 0, dtype=torch.int64), 10, "cpu", T=2)      # empty edge set is fine


def _vgpr_tool(*args, timeout=900):
    import subprocess
    import sys
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_reserved_vgprs.py"), *args], capture_output=True,
                          text=True, timeout=timeout)


def test_reserved_register_zone_is_untouched_by_compiler_code():
    """The stream kernels park in-flight global loads in fixed registers at the top of the register
    file (csrc/async_stage.h).  That is only sound if no compiler-generated instruction uses that
    zone.  The BUILD enforces it (csrc/Makefile runs the tool on the assembly of the compile that
    produced gemm.o / mtransform.o and on the linked library, and fails on a violation); here the
    same two checks run once more: kernels recompiled with the Makefile's own flags, and the
    disassembly of the library that ships."""
    r = _vgpr_tool()
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(": OK") == 10, r.stdout   # (4 GEMM instantiations + the M-transform) x (assembly, shipped library)


def test_reserved_register_check_is_part_of_the_build():
    mk = open(os.path.join(ROOT, "tm-gcn_amd", "csrc", "Makefile")).read()
    assert "check_reserved_vgprs.py" in mk and "--asm" in mk and "--lib" in mk and "-save-temps=obj" in mk


def test_reserved_register_checker_flags_violations():
    """Negative tests of the checker: synthetic assembly with each kind of violation (a reserved
    register in compiler code, also behind an early s_endpgm and inside a register range; a call;
    a missing / unlisted kernel), and the disassembly whitelist."""
    r = _vgpr_tool("--selftest", timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "WRONG" not in r.stdout and r.stdout.count("as expected") == 18, r.stdout


def test_reserved_register_checker_rejects_a_real_violating_kernel(tmp_path):
    """tests/vgpr_check/violates_reserved_zone.hip keeps ~230 accumulators live, so hipcc allocates
    far above v192 while its async_stage.h load targets v[192:195]: the tool must exit non-zero."""
    import subprocess
    out = tmp_path / "neg.s"
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tm-gcn_amd", "csrc"), "-S", "--cuda-device-only",
                           os.path.join(HERE, "vgpr_check", "violates_reserved_zone.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    r = _vgpr_tool("--asm", str(out), "--kernels", "neg_reserved_zone_kernel:192:1", timeout=120)
    assert r.returncode == 1 and "VIOLATIONS" in r.stdout, r.stdout + r.stderr
