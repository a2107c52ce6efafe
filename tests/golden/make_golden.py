"""Generate the golden fixtures tests/golden/*.npz by running the REAL reference.

Run in the build container only (needs /root/reference; the GPU box has neither the reference
nor this need — the committed .npz files travel instead):

    python tests/golden/make_golden.py

What it does: imports /root/reference/TensorGCN-master/embedding_help_functions.py ("ehf",
with an empty torchvision stub — ehf imports it but never uses it), builds small seeded inputs
with tmgcn_amd.synth, runs the reference classes, and stores inputs + outputs + gradients.
For G5 it also ast-extracts the preprocessing *functions* of read_data.py (a script with
hard-coded paths that cannot be imported) and runs them on a node-subsampled slice of the one
dataset the reference ships (data/chess), which pins tmgcn_amd.synth's restated pipeline.

Fixtures are data only (arrays): no reference source text is stored.
"""
import ast
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/TensorGCN-master"
sys.path.insert(0, ROOT)

for m in ("torchvision", "torchvision.datasets"):
    sys.modules.setdefault(m, types.ModuleType(m))
sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
sys.path.insert(0, REF)
import embedding_help_functions as ehf  # noqa: E402  (the real reference)

import tmgcn_amd.synth as synth  # noqa: E402

torch.set_num_threads(4)


def coo_arrays(mats):
    ks, rs, cs, vs = [], [], [], []
    for k, m in enumerate(mats):
        m = m.tocoo()
        ks.append(np.full(m.nnz, k, np.int32))
        rs.append(m.row.astype(np.int32))
        cs.append(m.col.astype(np.int32))
        vs.append(m.data.astype(np.float64))
    return np.concatenate(ks), np.concatenate(rs), np.concatenate(cs), np.concatenate(vs)


def ref_list(mats):
    """The reference's list-of-COO form: built WITHOUT explicit size, as ehf:564 does."""
    out = []
    for m in mats:
        m = m.tocoo()
        idx = torch.tensor(np.stack([m.row, m.col]), dtype=torch.long)
        out.append(torch.sparse.DoubleTensor(idx, torch.tensor(m.data, dtype=torch.float64)))
    return out


def graph_inputs(g, prefix=""):
    d = {}
    for name, mats in (("At", g.Ct), ("A", g.C)):
        k, r, c, v = coo_arrays(mats)
        d.update({f"{prefix}{name}_k": k, f"{prefix}{name}_i": r, f"{prefix}{name}_j": c, f"{prefix}{name}_v": v})
    d[prefix + "X"] = g.X
    d[prefix + "M"] = g.M
    d[prefix + "edges"] = g.edges
    d[prefix + "labels"] = g.labels
    return d


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def loss_and_grads(model, target, alpha=0.9):
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([alpha, 1.0 - alpha]))
    out = model()
    loss = crit(out, target)
    model.zero_grad()
    loss.backward()
    return out.detach().numpy(), float(loss), {n: p.grad.detach().numpy().copy() for n, p in model.named_parameters()}


# ------------------------------------------------------------------------------ G1
def g1():
    for T, N, F in ((5, 7, 2), (34, 200, 16), (5, 200, 2), (34, 7, 16)):
        g = synth.dynamic_graph(T, N, edges_per_slice=max(4, N // 2), seed=T * 1000 + N, no_diag=min(20, T), F0=F)
        torch.manual_seed(0)
        m = ehf.EmbeddingGCN(ref_list(g.Ct), torch.tensor(g.X), torch.tensor(g.edges), torch.tensor(g.M),
                             hidden_feat=[3, 2], condensed_W=True, use_Minv=False)
        save(f"g1_AtXt_T{T}_N{N}_F{F}", AtXt=m.AtXt.numpy(), **graph_inputs(g))


# ------------------------------------------------------------------------------ G2
def g2():
    g = synth.dynamic_graph(12, 60, 90, seed=2, no_diag=5)
    base = graph_inputs(g)
    tgt = torch.tensor(g.labels)
    for condensed in (True, False):
        torch.manual_seed(11)
        m = ehf.EmbeddingGCN(ref_list(g.Ct), torch.tensor(g.X), torch.tensor(g.edges), torch.tensor(g.M),
                             hidden_feat=[6, 2], condensed_W=condensed, use_Minv=False)
        W0, U0 = m.W.detach().numpy().copy(), m.U.detach().numpy().copy()
        out, loss, grads = loss_and_grads(m, tgt)
        save(f"g2_gcn_condensed{int(condensed)}", seed=11, W0=W0, U0=U0, logits=out, loss=loss,
             dW=grads["W"], dU=grads["U"], **base)
    # use_Minv=True runs in the reference only with all-fp32 inputs
    torch.manual_seed(12)
    At32 = [a.float() for a in ref_list(g.Ct)]
    m = ehf.EmbeddingGCN(At32, torch.tensor(g.X).float(), torch.tensor(g.edges), torch.tensor(g.M).float(),
                         hidden_feat=[6, 2], condensed_W=True, use_Minv=True)
    W0, U0 = m.W.detach().numpy().copy(), m.U.detach().numpy().copy()
    out, loss, grads = loss_and_grads(m, tgt)
    save("g2_gcn_minv_fp32", seed=12, W0=W0, U0=U0, logits=out, loss=loss, dW=grads["W"], dU=grads["U"], **base)


# ------------------------------------------------------------------------------ G3
def g3():
    g = synth.dynamic_graph(10, 50, 80, seed=3, no_diag=4)
    gv = synth.dynamic_graph(10, 50, 70, seed=33, no_diag=4)  # a "validation" window
    base = graph_inputs(g)
    base.update(graph_inputs(gv, prefix="val_"))
    tgt = torch.tensor(g.labels)
    branches = {"default": dict(), "twice": dict(apply_M_twice=True),
                "three": dict(apply_M_twice=True, apply_M_three_times=True)}
    for bname, kw in branches.items():
        for nl in ("relu", "leaky", "selu"):
            for condensed in ((True, False) if (bname == "twice" and nl == "selu") else (True,)):
                torch.manual_seed(21)
                m = ehf.EmbeddingGCN2(ref_list(g.Ct), torch.tensor(g.X), torch.tensor(g.edges), torch.tensor(g.M),
                                      hidden_feat=[6, 6, 2], condensed_W=condensed, use_Minv=False, nonlin2=nl, **kw)
                p0 = {n + "0": p.detach().numpy().copy() for n, p in m.named_parameters()}
                out, loss, grads = loss_and_grads(m, tgt)
                with torch.no_grad():  # validation-style call: layer 2 still uses the training At (ehf:343/348)
                    out_val = m(ref_list(gv.Ct), torch.tensor(gv.X), torch.tensor(gv.edges)).numpy()
                save(f"g3_gcn2_{bname}_{nl}_condensed{int(condensed)}", seed=21, logits=out, loss=loss,
                     logits_val=out_val, dW1=grads["W1"], dW2=grads["W2"], dU=grads["U"], **p0, **base)


# ------------------------------------------------------------------------------ G4
def g4():
    g = synth.dynamic_graph(9, 40, 60, seed=4, no_diag=3)
    # a SHORTER validation window (the baseline scripts train on 150 slices and validate on 25):
    # compute_AX zero-pads to the training T (ehf:469-473) and layer 2 runs over all of self.A
    gv = synth.dynamic_graph(4, 40, 50, seed=44, no_diag=3)
    base = graph_inputs(g)
    base.update(graph_inputs(gv, prefix="val_"))
    tgt = torch.tensor(g.labels)
    for hf, nl in (([6, 2], "relu"), ([6, 5, 2], "selu"), ([6, 5, 2], "leaky")):
        torch.manual_seed(31)
        m = ehf.EmbeddingKWGCN(ref_list(g.C), torch.tensor(g.X), torch.tensor(g.edges), hidden_feat=hf, nonlin2=nl)
        p0 = {n + "0": p.detach().numpy().copy() for n, p in m.named_parameters()}
        out, loss, grads = loss_and_grads(m, tgt)
        with torch.no_grad():
            out_val = m(ref_list(gv.C), torch.tensor(gv.X), torch.tensor(gv.edges)).numpy()
        save(f"g4_kwgcn_{len(hf) - 1}layer_{nl}", seed=31, logits=out, loss=loss, logits_val=out_val,
             **{"d" + n: v for n, v in grads.items()}, **p0, **base)


# ------------------------------------------------------------------------------ G5
def extract_functions(path, names, env):
    """Compile selected top-level function definitions of a reference script into `env`."""
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), env)
    return env


def g5():
    raw = np.loadtxt(os.path.join(REF, "data/chess/out.chess.csv"), comments="%")
    keep_nodes, TT = 300, 16
    dates = np.unique(raw[:, 3])[:TT]
    sel = (raw[:, 0] <= keep_nodes) & (raw[:, 1] <= keep_nodes) & np.isin(raw[:, 3], dates)
    data = raw[sel]
    N = keep_nodes
    t_idx = np.searchsorted(dates, data[:, 3])
    idx = torch.tensor(np.stack([t_idx, data[:, 0] - 1, data[:, 1] - 1]), dtype=torch.long)
    A = torch.sparse.DoubleTensor(idx, torch.ones(idx.shape[1], dtype=torch.double), torch.Size([TT, N, N])).coalesce()
    env = {"torch": torch, "np": np, "edge_life_window": 10, "no_diag": 20}
    extract_functions(os.path.join(REF, "read_data.py"),
                      {"func_make_symmetric", "func_edge_life", "func_laplacian_transformation", "func_MProduct"}, env)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        B = env["func_edge_life"](env["func_make_symmetric"](A, N, TT), N, TT)
        Cn = env["func_laplacian_transformation"](B, N, TT)
        # read_data.py:55-62 M: ones on 20 diagonals, row-normalised
        M = np.zeros((TT, TT))
        for i in range(20):
            np.fill_diagonal(M[i:, :TT - i], 1)
        M = M / M.sum(axis=1)[:, None]
        Ct = env["func_MProduct"](Cn, torch.tensor(M))
    Ai = A._indices().numpy()
    Ci, Cv = Cn._indices().numpy(), Cn._values().numpy()
    Ti, Tv = Ct._indices().numpy(), Ct._values().numpy()
    # then the model on top of it (chess has 3 label classes -1/0/1; use sign>=0 as the 2-class target)
    edges = Ai
    labels = (np.asarray(A._values().numpy()) > 0).astype(np.int64)
    lab_src = {(int(a), int(b), int(c)): int(w >= 0) for a, b, c, w in zip(t_idx, data[:, 0] - 1, data[:, 1] - 1, data[:, 2])}
    labels = np.array([lab_src[(int(a), int(b), int(c))] for a, b, c in edges.T], dtype=np.int64)
    At_list = [torch.sparse.DoubleTensor(torch.tensor(Ti[1:3, Ti[0] == k]), torch.tensor(Tv[Ti[0] == k])) for k in range(TT)]
    Xf = torch.zeros(TT, N, 2)
    Xf[:, :, 0] = torch.sparse.sum(A, 1).to_dense()
    Xf[:, :, 1] = torch.sparse.sum(A, 2).to_dense()
    torch.manual_seed(41)
    m = ehf.EmbeddingGCN2(At_list, Xf.double(), torch.tensor(edges), torch.tensor(M), hidden_feat=[6, 6, 2],
                          condensed_W=True, use_Minv=False, nonlin2="selu")
    p0 = {n + "0": p.detach().numpy().copy() for n, p in m.named_parameters()}
    out, loss, grads = loss_and_grads(m, torch.tensor(labels))
    save("g5_chess_gcn2", seed=41, T=TT, N=N, raw_k=Ai[0].astype(np.int32), raw_i=Ai[1].astype(np.int32),
         raw_j=Ai[2].astype(np.int32), C_k=Ci[0].astype(np.int32), C_i=Ci[1].astype(np.int32), C_j=Ci[2].astype(np.int32),
         C_v=Cv, At_k=Ti[0].astype(np.int32), At_i=Ti[1].astype(np.int32), At_j=Ti[2].astype(np.int32), At_v=Tv,
         M=M, X=Xf.double().numpy(), edges=edges, labels=labels, logits=out, loss=loss,
         dW1=grads["W1"], dW2=grads["W2"], dU=grads["U"], **p0)


# ------------------------------------------------------------------------------ G6
def g6():
    g = synth.dynamic_graph(10, 50, 80, seed=6, no_diag=4)
    base = graph_inputs(g)
    tgt = torch.tensor(g.labels)
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1]))
    for name, ctor in (
        ("gcn", lambda: ehf.EmbeddingGCN(ref_list(g.Ct), torch.tensor(g.X), torch.tensor(g.edges), torch.tensor(g.M),
                                          hidden_feat=[6, 2], condensed_W=True, use_Minv=False)),
        ("gcn2", lambda: ehf.EmbeddingGCN2(ref_list(g.Ct), torch.tensor(g.X), torch.tensor(g.edges), torch.tensor(g.M),
                                            hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False, nonlin2="selu")),
    ):
        torch.manual_seed(51)
        m = ctor()
        # scale the N(0,1) init down so that 10 SGD steps stay in a numerically tame regime
        opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
        losses = []
        for _ in range(10):
            opt.zero_grad()
            loss = crit(m(), tgt)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        final = {n + "_final": p.detach().numpy().copy() for n, p in m.named_parameters()}
        save(f"g6_sgd_{name}", seed=51, losses=np.array(losses), **final, **base)


# ------------------------------------------------------------------------------ G8
def g8():
    """EmbeddingGCN_reg (ehf:359-423): regression head, MSE loss as in test_graph_SEIR.py."""
    g = synth.dynamic_graph(9, 40, 60, seed=8, no_diag=3)
    base = graph_inputs(g)
    y = torch.randn(9, 40, generator=torch.Generator().manual_seed(80))
    for condensed in (True, False):
        torch.manual_seed(61)
        m = ehf.EmbeddingGCN_reg(ref_list(g.Ct), torch.tensor(g.X), torch.tensor(g.M), hidden_feat=[6],
                                 condensed_W=condensed, use_Minv=False)
        p0 = {n.replace(".", "_") + "0": p.detach().numpy().copy() for n, p in m.named_parameters()}
        out = m()
        loss = torch.nn.MSELoss()(out, y)
        m.zero_grad()
        loss.backward()
        grads = {"d" + n.replace(".", "_"): p.grad.detach().numpy().copy() for n, p in m.named_parameters()}
        save(f"g8_gcn_reg_condensed{int(condensed)}", seed=61, out=out.detach().numpy(), y=y.numpy(), loss=float(loss),
             **p0, **grads, **base)


# ------------------------------------------------------------------------------ G10
def _pack_sym(S, N):
    """A coalesced, pattern-symmetric sparse [T,N,N] tensor -> compact arrays: entries with row <= col only
    (per-(slice,row) counts, uint16 columns, the reference's fp64 values ROUNDED TO fp32) plus fp64 slice sums
    of the unrounded values.  Asserted here, on the reference's real output: the pattern is symmetric and
    the fp32-rounded values are exactly symmetric, so tests/_util.unpack_sym() restores every entry."""
    idx, v = S._indices().numpy(), S._values().numpy()
    T = int(S.shape[0])
    key = (idx[0].astype(np.int64) * N + idx[1]) * N + idx[2]
    assert (np.diff(key) > 0).all(), "coalesced tensors are sorted"
    keyT = (idx[0].astype(np.int64) * N + idx[2]) * N + idx[1]
    o = np.argsort(keyT, kind="stable")
    v32 = v.astype(np.float32)
    assert (keyT[o] == key).all() and (v32[o] == v32).all()
    up = idx[1] <= idx[2]
    counts = np.bincount(idx[0][up] * N + idx[1][up], minlength=T * N)
    assert counts.max() < 2 ** 16 and N < 2 ** 16
    sums = np.bincount(idx[0], weights=v, minlength=T)
    return {"cnt": counts.astype(np.uint16), "col": idx[2][up].astype(np.uint16), "val": v32[up], "nnz": np.int64(len(v)),
            "slice_sum": sums}


def _chess_reference_setup(tag):
    """The reference's own preprocessing of data/chess (read_data.py 'Chess' settings) and experiment_chess_our.py's
    inputs, as a dict of the local names g10() and g11() use."""
    import time
    import warnings
    warnings.simplefilter("ignore")
    raw = np.loadtxt(os.path.join(REF, "data/chess/out.chess.csv"), comments="%")
    dates = np.unique(raw[:, 3])                                        # read_data.py:46-47
    TT, N = len(dates), int(max(raw[:, 0].max(), raw[:, 1].max()))      # :51
    S_train, S_val, S_test = 80, 10, 10                                  # :36-38, experiment_chess_our.py:32
    T = S_train
    t_idx = np.searchsorted(dates, raw[:, 3])                           # :74-83 (one slice per distinct date)
    tidx = torch.tensor(np.stack([t_idx, raw[:, 0] - 1, raw[:, 1] - 1]), dtype=torch.long)
    A = torch.sparse.DoubleTensor(tidx, torch.ones(tidx.shape[1], dtype=torch.double), torch.Size([TT, N, N])).coalesce()
    A_labels = torch.sparse.DoubleTensor(tidx, torch.tensor(raw[:, 2]), torch.Size([TT, N, N])).coalesce()   # :85-86
    env = {"torch": torch, "np": np, "edge_life_window": 10, "no_diag": 20}
    extract_functions(os.path.join(REF, "read_data.py"),
                      {"func_make_symmetric", "func_edge_life", "func_laplacian_transformation", "func_create_sparse",
                       "func_MProduct"}, env)
    M = np.zeros((T, T))                                                 # :55-62
    for i in range(20):
        np.fill_diagonal(M[i:, :T - i], 1)
    M = M / M.sum(axis=1)[:, None]
    t0 = time.time()
    B = env["func_edge_life"](env["func_make_symmetric"](A, N, TT), N, TT)
    Cn = env["func_laplacian_transformation"](B, N, TT)
    C_train = env["func_create_sparse"](Cn, N, TT, T, 0, T)              # :186-188
    C_val = env["func_create_sparse"](Cn, N, TT, T, S_val, T + S_val)
    Ct_train = env["func_MProduct"](C_train, torch.tensor(M))            # :225-227
    Ct_val = env["func_MProduct"](C_val, torch.tensor(M))
    print(f"{tag}: reference preprocessing {time.time() - t0:.0f} s; nnz A {A._nnz()}, C {Cn._nnz()}, Ct_train {Ct_train._nnz()}")

    def slices(S):       # experiment_chess_our.py:54-57 — per-slice matrices WITHOUT an explicit size (ehf:564)
        i, v = S._indices(), S._values()
        return [torch.sparse.DoubleTensor(i[1:3, i[0] == j], v[i[0] == j]) for j in range(T)]

    # experiment_chess_our.py:66-92 (features, edge sets, 3-class targets)
    li, lv = A_labels._indices(), A_labels._values()
    A1 = torch.sparse.FloatTensor(li, torch.ones(lv.shape), torch.Size([TT, N, N])).coalesce()   # :51 — the 0/1 pattern
    X = torch.zeros(TT, N, 2)
    X[:, :, 0] = torch.sparse.sum(A1, 1).to_dense()
    X[:, :, 1] = torch.sparse.sum(A1, 2).to_dense()
    X_train, X_val = X[0:S_train].double(), X[S_val:S_train + S_val].double()
    edges_train = li[:, li[0] < S_train]
    target_train = (torch.sign(lv[li[0] < S_train]) + 1).long()
    sv = (li[0] >= S_val) & (li[0] < S_train + S_val)
    edges_val = li[:, sv].clone()
    edges_val[0] -= S_val
    eval_val = edges_val[0] >= S_train - S_val
    At_train, At_val = slices(Ct_train), slices(Ct_val)
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([.33, .33, .33]))     # experiment_chess_our.py:23, 99
    return dict(locals())


def g10():
    """The reference's OWN preprocessing on the whole chess data set it ships — all 7 301 players, all 100
    monthly slices, read_data.py's 'Chess' settings (edge life 10, 20 diagonals, symmetric; 80 / 10 / 10
    train / val / test slices) — then experiment_chess_our.py's models on the 80-slice training block
    (T = 80 > no_diag = 20: the band of M is truncated, which fixture G5 at T = 16 never was)."""
    c = _chess_reference_setup("g10")
    (TT, N, S_train, S_val, S_test, T, M, t_idx, raw, Cn, Ct_train, Ct_val, X, X_train, X_val, li, lv, sv, edges_train, target_train,
     edges_val, eval_val, At_train, At_val, crit) = (c[k] for k in (
        "TT", "N", "S_train", "S_val", "S_test", "T", "M", "t_idx", "raw", "Cn", "Ct_train", "Ct_val", "X", "X_train", "X_val", "li", "lv",
        "sv", "edges_train", "target_train", "edges_val", "eval_val", "At_train", "At_val", "crit"))
    out = {}
    models = {
        "gcn": lambda: ehf.EmbeddingGCN(At_train, X_train, edges_train, torch.tensor(M), hidden_feat=[6, 3],
                                        condensed_W=True, use_Minv=False),                      # :94 (no_layers == 1)
        "gcn2": lambda: ehf.EmbeddingGCN2(At_train, X_train, edges_train, torch.tensor(M), hidden_feat=[6, 6, 3],
                                          condensed_W=True, use_Minv=False, nonlin2="selu"),    # :92 (no_layers == 2)
        "gcn2_twice": lambda: ehf.EmbeddingGCN2(At_train, X_train, edges_train, torch.tensor(M), hidden_feat=[6, 6, 3],
                                                condensed_W=True, use_Minv=False, nonlin2="selu", apply_M_twice=True),
    }
    for name, ctor in models.items():
        torch.manual_seed(71)
        m = ctor()
        for n, p in m.named_parameters():
            out[f"{name}_{n}0"] = p.detach().numpy().copy()
        logits = m()
        loss = crit(logits, target_train)
        m.zero_grad()
        loss.backward()
        out[f"{name}_logits"] = logits.detach().numpy()
        out[f"{name}_loss"] = float(loss)
        for n, p in m.named_parameters():
            out[f"{name}_d{n}"] = p.grad.detach().numpy().copy()
        if name != "gcn2_twice":
            with torch.no_grad():    # :112-115 — the validation call; the script scores the last S_val slices only
                lv_ = m(At_val, X_val, edges_val)
                out[f"{name}_logits_val_eval"] = lv_[eval_val].numpy()
                out[f"{name}_loss_val"] = float(crit(lv_[eval_val], (torch.sign(lv[sv]) + 1).long()[eval_val]))
        print(f"g10: {name} loss {float(loss):.6f}")
    # the scripts' training loop itself (experiment_chess_our.py:97-108): 6 SGD steps (lr .01, momentum .9) of the 2-layer model
    torch.manual_seed(71)
    m = models["gcn2"]()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
    traj = []
    for _ in range(6):
        opt.zero_grad()
        loss = crit(m(), target_train)
        loss.backward()
        opt.step()
        traj.append(float(loss))
    out["gcn2_sgd_losses"] = np.array(traj)
    for n, p in m.named_parameters():
        out[f"gcn2_sgd_{n}_final"] = p.detach().numpy().copy()
    print("g10: gcn2 SGD losses", traj)
    # experiment_chess_baseline.py:47-90, 101-104 — the baseline without the M-product on the un-transformed C: training
    # on slices 0..79, validation on the SHORTER window 80..89 (compute_AX zero-pads to the training T, ehf:469-473)
    Ci, Cv = Cn._indices(), Cn._values()
    C_train_l = [torch.sparse.DoubleTensor(Ci[1:3, Ci[0] == j], Cv[Ci[0] == j]) for j in range(S_train)]
    C_val_l = [torch.sparse.DoubleTensor(Ci[1:3, Ci[0] == j], Cv[Ci[0] == j]) for j in range(S_train, S_train + S_val)]
    Xb_val = X[S_train:S_train + S_val].double()
    svb = (li[0] >= S_train) & (li[0] < S_train + S_val)
    edges_val_b = li[:, svb].clone()
    edges_val_b[0] -= S_train
    for name, hf in (("kw1", [6, 3]), ("kw2", [6, 6, 3])):
        torch.manual_seed(71)
        m = ehf.EmbeddingKWGCN(C_train_l, X_train, edges_train, hf, nonlin2="selu")
        for n, p in m.named_parameters():
            out[f"{name}_{n}0"] = p.detach().numpy().copy()
        logits = m()
        loss = crit(logits, target_train)
        m.zero_grad()
        loss.backward()
        out[f"{name}_logits"] = logits.detach().numpy()
        out[f"{name}_loss"] = float(loss)
        for n, p in m.named_parameters():
            out[f"{name}_d{n}"] = p.grad.detach().numpy().copy()
        with torch.no_grad():
            out[f"{name}_logits_val"] = m(C_val_l, Xb_val, edges_val_b).numpy()
        print(f"g10: {name} loss {float(loss):.6f}")
    packed = {}
    for nm, S in (("C", Cn), ("Ct", Ct_train)):
        packed.update({f"{nm}_{k}": v for k, v in _pack_sym(S, N).items()})
    # Ct_val is not stored (the device pipeline and the CPU restatement rebuild it from the raw edges / from C);
    # its slice sums are, so that a test can tell a wrong validation block from a wrong model
    packed["Ct_val_slice_sum"] = np.bincount(Ct_val._indices()[0].numpy(), weights=Ct_val._values().numpy(), minlength=T)
    packed["Ct_val_nnz"] = np.int64(Ct_val._nnz())
    save("g10_chess_full", seed=71, TT=TT, N=N, S_train=S_train, S_val=S_val, S_test=S_test, M=M,
         raw_k=t_idx.astype(np.uint8), raw_i=(raw[:, 0] - 1).astype(np.uint16), raw_j=(raw[:, 1] - 1).astype(np.uint16),
         raw_label=raw[:, 2].astype(np.int8), **packed, **out)


def g11():
    """Long-horizon parity: experiment_chess_our.py's training loop (:108-123 — SGD lr .01 momentum .9, class-weighted CE,
    the 2-layer model) run for 300 epochs on the full chess data from G10's seed with the REAL ehf.EmbeddingGCN2: the loss
    of every epoch, the script's train / validation accuracy and validation loss at epochs 0 / 100 / 200 / 299 (its
    `ep % 100 == 0` block, :117-123, plus the last epoch), the argmax class counts there, the final W1 / W2 / U, and the
    parameters + momentum buffers in front of epochs 100 / 200 / 280 (so that the CPU oracle can be pinned on the late
    part of the trajectory without walking all of it).
    Inputs are G10's (tests/golden/g10_chess_full.npz holds the raw edges); this fixture stores outputs only."""
    import time
    c = _chess_reference_setup("g11")
    torch.manual_seed(71)
    m = ehf.EmbeddingGCN2(c["At_train"], c["X_train"], c["edges_train"], torch.tensor(c["M"]), hidden_feat=[6, 6, 3],
                          condensed_W=True, use_Minv=False, nonlin2="selu")
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
    crit, target_train = c["crit"], c["target_train"]
    target_val = (torch.sign(c["lv"][c["sv"]]) + 1).long()
    ev = c["eval_val"]
    n_ep = 300
    losses, marks, ckpt = [], [], {}
    t0 = time.time()
    for ep in range(n_ep):
        if ep in (100, 200, 280):        # the state BEFORE epoch `ep`: parameters and SGD momentum buffers (a checker can resume here)
            for n, q in m.named_parameters():
                ckpt[f"ckpt{ep}_{n}"] = q.detach().numpy().copy()
                ckpt[f"ckpt{ep}_mom_{n}"] = opt.state[q]["momentum_buffer"].numpy().copy()
        opt.zero_grad()
        out = m()
        loss = crit(out, target_train)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        if ep % 100 == 0 or ep == n_ep - 1:
            with torch.no_grad():
                guess = torch.argmax(out, dim=1)
                acc_train = int(torch.sum(guess == target_train)) / len(guess)
                out_val = m(c["At_val"], c["X_val"], c["edges_val"])        # AFTER the step, as the script does (:117)
                gv = torch.argmax(out_val, dim=1)
                acc_val = int(torch.sum(gv[ev] == target_val[ev])) / int(ev.sum())
                loss_val = float(crit(out_val[ev], target_val[ev]))
                marks.append([ep, acc_train, acc_val, loss_val] + torch.bincount(guess, minlength=3).tolist()
                             + torch.bincount(gv[ev], minlength=3).tolist())
            print(f"g11: ep {ep} loss {float(loss):.6f} acc_train {acc_train:.4f} acc_val {acc_val:.4f} loss_val {loss_val:.6f} "
                  f"({time.time() - t0:.0f} s)")
    save("g11_chess_train300", seed=71, epochs=n_ep, lr=0.01, momentum=0.9, losses=np.array(losses),
         marks=np.array(marks, dtype=np.float64),
         marks_columns=np.array(["epoch", "acc_train", "acc_val", "loss_val", "train_argmax_0", "train_argmax_1", "train_argmax_2",
                                 "val_argmax_0", "val_argmax_1", "val_argmax_2"]),
         **ckpt, **{f"{n}_final": p.detach().numpy().copy() for n, p in m.named_parameters()})


if __name__ == "__main__":
    which = sys.argv[1:]
    for name, fn in (("g1", g1), ("g2", g2), ("g3", g3), ("g4", g4), ("g5", g5), ("g6", g6), ("g8", g8), ("g10", g10), ("g11", g11)):
        if not which or name in which:
            fn()
