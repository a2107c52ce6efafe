"""Helpers shared by the tests: fixture loading and tolerances."""
import ctypes as C
import glob
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")

# Stated tolerance (SURVEY §8c / DESIGN.md §5): fp32 build vs the reference's fp64->fp32 path.
REL_TOL = 1e-5


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def coo_list(d, name, T, N, dtype=torch.float64, prefix=""):
    """Fixture COO arrays -> the reference's list of per-slice sparse tensors."""
    k, i, j, v = (d[f"{prefix}{name}_{s}"] for s in "kijv")
    out = []
    for t in range(T):
        m = k == t
        idx = torch.from_numpy(np.stack([i[m], j[m]]).astype(np.int64))
        out.append(torch.sparse_coo_tensor(idx, torch.from_numpy(v[m]).to(dtype), (N, N)))
    return out


def max_rel_err(got, ref):
    got = torch.as_tensor(got).double().cpu()
    ref = torch.as_tensor(ref).double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = max(float(ref.abs().max()), 1e-30)
    return float((got - ref).abs().max()) / scale


def record_tolerance(what, err, tol, kind="max|Δ|/max|ref|"):
    """Every assertion whose bound is LOOSER than the stated 1e-5 (SGD trajectories, all-fp32
    fixtures, MAP/MRR, bf16 weights, identities) appends its MEASURED error to
    gpurun_out/tolerance_record.jsonl — one JSON line per assertion, written from whichever process
    made it (the multi-process tests assert inside their workers) — so that a drift toward a bound
    shows in the record long before the assertion trips.  tools/tolerance_summary.py condenses the
    file; a copy of the summary is committed under profiles/."""
    if tol <= REL_TOL:
        return
    import json
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        rec = {"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "what": what, "kind": kind,
               "measured": float(err), "bound": float(tol), "used": float(err) / float(tol) if tol else None}
        with open(os.path.join(out, "tolerance_record.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def assert_close(got, ref, tol=REL_TOL, what=""):
    err = max_rel_err(got, ref)
    record_tolerance(what, err, tol)
    assert err <= tol, f"{what}: max|Δ|/max|ref| = {err:.3e} > {tol:.1e}"


def load_c_oracle():
    from oracle import c_ref
    return c_ref.load()


def cptr(t):
    return C.c_void_p(t.data_ptr())


def free_port():
    """A TCP port that is free on loopback right now (rendezvous of the multi-process tests: fixed
    ports collide when two suites share a box)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def unpack_sym(d, name, T, N):
    """Inverse of tests/golden/make_golden.py:_pack_sym — the compact upper-triangle form of fixture G10
    back to every stored entry of the reference's coalesced sparse tensor: (slice, row, col, value) arrays
    sorted by (slice, row, col), values as fp32 (the reference's fp64 values rounded once; the generator
    asserted that the pattern is symmetric and the rounded values exactly so)."""
    cnt = d[name + "_cnt"].astype(np.int64)
    col = d[name + "_col"].astype(np.int64)
    val = d[name + "_val"]
    rows = np.repeat(np.arange(T * N, dtype=np.int64), cnt)          # slice*N + row of every upper entry
    k, i = rows // N, rows % N
    off = i != col
    K = np.concatenate((k, k[off]))
    I = np.concatenate((i, col[off]))
    J = np.concatenate((col, i[off]))
    V = np.concatenate((val, val[off]))
    order = np.argsort((K * N + I) * N + J, kind="stable")
    assert len(order) == int(d[name + "_nnz"])
    return K[order], I[order], J[order], V[order]


def head_loss_fp64(Z, W, U, edges, target, weight, N):
    """The scripts' head + criterion in fp64 with stock torch ops + autograd, on whatever device Z lives:
    logits = [Y[src], Y[dst]]·U with Y = Z (or Z·W: the folded 1-layer model, ehf:222), ehf:228-232;
    loss = nn.CrossEntropyLoss(weight)(logits, target) (experiment_reddit_our_link_prediction.py:69, 79).
    The checker of tests/test_gpu_head_loss.py; itself pinned on fixture G2's logits / loss / dW / dU (the real reference's
    numbers) by tests/test_oracle_golden.py::test_head_loss_fp64_restatement_is_pinned_on_g2.
    Returns (logits, loss, dZ, dU, dW or None)."""
    Zd = Z.double().requires_grad_(True)
    Ud = U.double().requires_grad_(True)
    Wd = W.double().requires_grad_(True) if W is not None else None
    Y = (Zd @ Wd if W is not None else Zd).reshape(-1, U.shape[0] // 2)
    e = edges.to(Z.device)
    logits = torch.cat((Y[e[0] * N + e[1]], Y[e[0] * N + e[2]]), dim=1) @ Ud
    loss = torch.nn.functional.cross_entropy(logits, target, weight=weight.double())
    loss.backward()
    return logits.detach(), loss.detach(), Zd.grad, Ud.grad, (Wd.grad if W is not None else None)
