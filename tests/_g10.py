"""Fixture G10 (tests/golden/g10_chess_full.npz): the reference's own preprocessing and models on the WHOLE
chess data set it ships (7 301 players, 100 monthly slices, 80 / 10 / 10 train / val / test slices —
read_data.py's 'Chess' settings and experiment_chess_our.py's split).  This module turns the fixture's raw
rows into what the scripts derive from them (experiment_chess_our.py:47, 66-85): the edge-multiplicity
tensor's degree features, the labelled edge sets and the 3-class targets.  Shared by the CPU oracle test and
the GPU tests; numpy only."""
import numpy as np

from _util import golden, unpack_sym


class G10:
    def __init__(self):
        d = self.d = golden("g10_chess_full")
        self.TT, self.N = int(d["TT"]), int(d["N"])
        self.S_train, self.S_val, self.S_test = int(d["S_train"]), int(d["S_val"]), int(d["S_test"])
        self.T = self.S_train
        self.M = d["M"]
        k, i, j = d["raw_k"].astype(np.int64), d["raw_i"].astype(np.int64), d["raw_j"].astype(np.int64)
        self.raw = (k, i, j)
        N, TT = self.N, self.TT
        # A_labels = sparse(tensor_idx, labels).coalesce(): duplicate games between the same pair in the
        # same month SUM their labels; A = ones at the same pattern (experiment_chess_our.py:44, 51)
        key = (k * N + i) * N + j
        uk, inv = np.unique(key, return_inverse=True)
        lab = np.bincount(inv, weights=d["raw_label"].astype(np.float64), minlength=len(uk))
        ek, ei, ej = uk // (N * N), (uk // N) % N, uk % N
        self.edges_all = np.stack([ek, ei, ej])
        self.target_all = (np.sign(lab) + 1).astype(np.int64)            # 0 black win, 1 draw, 2 white win (:75)
        # node features (:60-65): X[t, n, 0] = Σ_i A[t, i, n], X[t, n, 1] = Σ_j A[t, n, j], A = 0/1 pattern
        X = np.zeros((TT, N, 2), np.float32)
        np.add.at(X[:, :, 0], (ek, ej), 1.0)
        np.add.at(X[:, :, 1], (ek, ei), 1.0)
        self.X = X.astype(np.float64)
        tr = ek < self.S_train
        self.edges_train, self.target_train = self.edges_all[:, tr], self.target_all[tr]
        va = (ek >= self.S_val) & (ek < self.S_train + self.S_val)      # same_block_size split (:78-83)
        self.edges_val = self.edges_all[:, va].copy()
        self.edges_val[0] -= self.S_val
        self.target_val = self.target_all[va]
        self.eval_val = self.edges_val[0] >= self.S_train - self.S_val
        self.X_train, self.X_val = self.X[:self.S_train], self.X[self.S_val:self.S_train + self.S_val]
        # the baseline's split (experiment_chess_baseline.py:64-81): validation = the S_val slices AFTER the training block
        vb = (ek >= self.S_train) & (ek < self.S_train + self.S_val)
        self.edges_val_b = self.edges_all[:, vb].copy()
        self.edges_val_b[0] -= self.S_train
        self.X_val_b = self.X[self.S_train:self.S_train + self.S_val]
        self.class_weights = np.array([.33, .33, .33], np.float32)      # experiment_chess_our.py:23

    def C(self):
        """(k, i, j, v) of the reference's normalised adjacency over all TT slices."""
        return unpack_sym(self.d, "C", self.TT, self.N)

    def Ct(self):
        """(k, i, j, v) of the reference's M-product of the 80-slice training block."""
        return unpack_sym(self.d, "Ct", self.T, self.N)

    @staticmethod
    def csr_arrays(k, i, j, T, N):
        """rowptr[T*N+1] (int64) of entries sorted by (slice, row, col)."""
        counts = np.bincount(k * N + i, minlength=T * N)
        rowptr = np.zeros(T * N + 1, np.int64)
        np.cumsum(counts, out=rowptr[1:])
        return rowptr
