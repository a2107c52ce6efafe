"""Metrics of the reference's experiment scripts, computed where the logits live.

    compute_f1        ehf.compute_f1:530-538        precision / recall / F1, class 0 = positive
    compute_MAP_MRR   ehf.compute_MAP_MRR:714-729   per-slice MAP (get_MAP:704-711, sklearn's
                                                    average_precision_score with pos_label=0) and
                                                    MRR (get_MRR:684-702, get_row_MRR:669-681),
                                                    weighted by the slices' share of the edges

The reference moves everything to numpy and scatters each slice into a dense rows×cols matrix
(ehf:692-693).  Its MRR has two quirks that are reproduced, not fixed, because parity is the
contract: (1) it ranks the RAW class-0 logit (`do_softmax=False`, ehf:726), so negative logits rank
below the matrix's zeros; (2) every unlabelled cell of the dense matrix has true_matrix == 0 and
therefore counts as an "existing" edge (ehf:670), contributing 1/rank of its zero.  Both have a
closed form — the zeros of a row occupy one contiguous block of ranks, so their contribution is a
difference of harmonic numbers — which lets the whole thing run on sorted segments without the
dense matrices.  Pure tensor code (works on CPU tensors too; fp64 like the reference).
"""
from __future__ import annotations

import torch


def compute_f1(guess: torch.Tensor, target: torch.Tensor):
    """ehf:530-538.  Returns (precision, recall, f1) as fp64 scalars; class 0 is the positive class."""
    tp = ((guess == 0) & (target == 0)).sum(dtype=torch.float64)
    fp = ((guess == 0) & (target != 0)).sum(dtype=torch.float64)
    fn = ((guess != 0) & (target == 0)).sum(dtype=torch.float64)
    precision = tp / (tp + fp)
    recall = tp / (tp + fn)
    return precision, recall, 2 * (precision * recall) / (precision + recall)


def _average_precision_pos0(score: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """sklearn.metrics.average_precision_score(target, score, pos_label=0): step-wise area under
    the precision-recall curve with one threshold per distinct score."""
    order = torch.argsort(score, descending=True, stable=True)
    s = score[order]
    pos = (target[order] == 0).to(torch.float64)
    tp = torch.cumsum(pos, 0)
    n = torch.arange(1, s.numel() + 1, device=s.device, dtype=torch.float64)
    last = torch.ones_like(pos, dtype=torch.bool)
    last[:-1] = s[1:] != s[:-1]            # last element of every run of equal scores
    P = tp[last] / n[last]
    R = tp[last] / tp[-1]
    dR = R - torch.cat((torch.zeros(1, device=R.device, dtype=R.dtype), R[:-1]))
    return (dR * P).sum()


def _mrr_slice(value: torch.Tensor, true: torch.Tensor, row: torch.Tensor, col: torch.Tensor) -> torch.Tensor:
    """get_MRR (ehf:684-702) for one slice without the dense matrices."""
    dev = value.device
    n_cols = int(col.max()) + 1
    # coo_matrix(...).toarray() sums duplicate (row, col) cells, for the scores and for the classes
    key = row * n_cols + col
    uk, inv = torch.unique(key, return_inverse=True)
    v = torch.zeros(uk.numel(), dtype=torch.float64, device=dev).index_add_(0, inv, value.double())
    t = torch.zeros(uk.numel(), dtype=torch.float64, device=dev).index_add_(0, inv, true.double())
    r = uk // n_cols
    # sort cells by (row, value descending)
    o1 = torch.argsort(v, descending=True, stable=True)
    o2 = torch.argsort(r[o1], stable=True)
    o = o1[o2]
    v, t, r = v[o], t[o], r[o]
    rows, counts = torch.unique_consecutive(r, return_counts=True)
    start = torch.cumsum(counts, 0) - counts
    seg = torch.repeat_interleave(torch.arange(rows.numel(), device=dev), counts)
    pos_in_row = torch.arange(v.numel(), device=dev) - start[seg]          # 0-based, by descending value
    n_pos = torch.zeros(rows.numel(), dtype=torch.int64, device=dev).index_add_(0, seg, (v > 0).long())
    zeros_block = n_cols - counts                                           # unlabelled cells of the row (value 0)
    # rank of a labelled cell: positives first, then the block of zeros, then the negative values
    rank = torch.where(v > 0, pos_in_row + 1, pos_in_row + 1 + zeros_block[seg]).double()
    existing = t == 0
    s_lab = torch.zeros(rows.numel(), dtype=torch.float64, device=dev).index_add_(0, seg, torch.where(existing, 1.0 / rank, torch.zeros_like(rank)))
    n_lab = torch.zeros(rows.numel(), dtype=torch.float64, device=dev).index_add_(0, seg, existing.double())
    H = torch.cat((torch.zeros(1, dtype=torch.float64, device=dev),
                   torch.cumsum(1.0 / torch.arange(1, n_cols + 1, device=dev, dtype=torch.float64), 0)))
    s_zero = H[n_pos + zeros_block] - H[n_pos]                             # Σ 1/rank over the zeros' block
    row_mrr = (s_lab + s_zero) / (n_lab + zeros_block.double())
    has_one = torch.zeros(rows.numel(), dtype=torch.float64, device=dev).index_add_(0, seg, (t == 1).double()) > 0
    return row_mrr[has_one].mean()                                          # rows that contain a class-1 cell (ehf:697)


def get_MAP(predictions: torch.Tensor, true_classes: torch.Tensor, do_softmax: bool = True) -> torch.Tensor:
    """ehf:704-711.  Average precision of class 0 over one edge set; ``predictions`` are [E,2]
    logits (do_softmax) or ready class-0 scores [E]."""
    score = torch.softmax(predictions, dim=1)[:, 0] if do_softmax else predictions
    return _average_precision_pos0(score, true_classes.to(score.device))


def get_MRR(predictions: torch.Tensor, true_classes: torch.Tensor, adj: torch.Tensor, do_softmax: bool = True) -> torch.Tensor:
    """ehf:684-702.  Mean over the rows of one slice of the mean reciprocal rank of its "existing"
    cells; ``adj`` = [2,E] (src, dst) of the scored edges."""
    score = torch.softmax(predictions, dim=1)[:, 0] if do_softmax else predictions[:, 0]
    adj = adj.to(score.device)
    return _mrr_slice(score, true_classes.to(score.device), adj[0], adj[1])


def get_row_MRR(probs, true_classes) -> torch.Tensor:
    """ehf:669-681.  One dense row: mean of 1/rank over the entries with class 0, ranks by
    descending score."""
    probs, true_classes = torch.as_tensor(probs), torch.as_tensor(true_classes)
    order = torch.argsort(probs, descending=True)
    ranks = torch.arange(1, probs.numel() + 1, dtype=torch.float64, device=probs.device)[(true_classes == 0)[order]]
    return (1.0 / ranks).sum() / ranks.numel()


def compute_MAP_MRR(output: torch.Tensor, target: torch.Tensor, edges: torch.Tensor, do_softmax: bool = True):
    """ehf:714-729.  output [E,2] logits, target [E], edges [3,E] (slice, src, dst).  Returns (MAP, MRR)."""
    edges = edges.to(output.device)
    target = target.to(output.device)
    MAP = torch.zeros((), dtype=torch.float64, device=output.device)
    MRR = torch.zeros((), dtype=torch.float64, device=output.device)
    total = edges.shape[1]
    for k in torch.unique(edges[0]).tolist():
        m = edges[0] == k
        w = m.sum().double() / total
        pred, tru = output[m], target[m]
        MAP = MAP + get_MAP(pred, tru, True) * w
        MRR = MRR + get_MRR(pred, tru, edges[1:3, m], False) * w
    return MAP, MRR
