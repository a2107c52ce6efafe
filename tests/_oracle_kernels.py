"""A CPU stand-in for tmgcn_amd.ops.kernels built from the ORACLE, used ONLY by the gloo tests
of the sharding logic (tests may use the oracle; the product never does)."""
import torch

from oracle import tmgcn_oracle as orc


def _perm(T, tl):
    """storage position -> logical row for the group-interleaved layout (include/tmgcn.h)."""
    k = torch.arange(T)
    pos = (k % tl) * (T // tl) + k // tl
    inv = torch.empty_like(pos)
    inv[pos] = k
    return pos, inv


def _view_coo(A):
    """list-of-COO (fp64) of a BatchedCSR or of one of its slice views (shared col/val storage)."""
    out = []
    for k in range(A.T):
        rp = A.rowptr[k * A.N:(k + 1) * A.N + 1]
        a, b = int(rp[0]), int(rp[-1])
        rows = torch.repeat_interleave(torch.arange(A.N), rp[1:] - rp[:-1])
        idx = torch.stack([rows, A.col[a:b].long()])
        out.append(torch.sparse_coo_tensor(idx, A.val[a:b].double(), (A.N, A.N)))
    return out


class OracleKernels:
    name = "oracle"

    def mtransform(self, op, X, transpose=False, row_off=0, col_off=0, T_out=None, x_group_rows=0, y_group_rows=0):
        T_in = X.shape[0]
        T_out = T_in if T_out is None else T_out
        if x_group_rows:
            pos, _ = _perm(T_in, x_group_rows)
            X = X[pos]  # logical row k lives at storage row pos[k]
        M = op.M64.t() if transpose else op.M64
        Mw = M[row_off:row_off + T_out, col_off:col_off + T_in]
        Y = torch.matmul(Mw, X.double().reshape(T_in, -1)).reshape((T_out,) + tuple(X.shape[1:])).float()
        if y_group_rows:
            pos, _ = _perm(T_out, y_group_rows)
            out = torch.empty_like(Y)
            out[pos] = Y
            Y = out
        return Y

    def mtransform_out(self, op, X, Y, transpose=False, row_off=0, col_off=0, x_group_rows=0, y_group_rows=0, tag=None):
        """Column-window form (tmgcn_mtransform_ld_f32): X / Y may be strided views; written into Y."""
        Y.copy_(self.mtransform(op, X.contiguous(), transpose, row_off, col_off, Y.shape[0], x_group_rows, y_group_rows))
        return Y

    def spmm(self, A, X, tag=None):
        return orc.slice_spmm(_view_coo(A), X.double())

    def spmm_gemm_supported(self, K, Nf):
        return (K % 8 == 0 and 16 <= K <= 128 and Nf <= 128) or (K in (1, 2, 3, 4, 6, 8) and Nf <= 16)  # the device rule

    def spmm_gemm(self, A, X, W, trans_w=False, act=None, want_ax=False, want_pre=False, tag=None, out=None, grid_reserve=0):
        lists = _view_coo(A)
        AX = orc.slice_spmm(lists, X.double())
        Y, pre = self.gemm(AX, W, trans_w=trans_w, act=act, want_pre=True)
        if out is not None:
            out[0].copy_(Y)
            if out[1] is not None:
                out[1].copy_(AX)
            if out[2] is not None and pre is not None:
                out[2].copy_(pre)
            return out
        return Y, (AX if want_ax else None), (pre if want_pre else None)

    def gemm(self, A, W, trans_w=False, act=None, want_pre=False, algo=None):
        Wd = W.double().transpose(-1, -2) if trans_w else W.double()     # W may be stored in bf16
        pre = torch.matmul(A.double(), Wd).float()
        Y = orc.ACTS[act](pre) if act else pre
        return (Y, pre if act else None) if want_pre else Y

    def gemm_dw(self, A, dY, per_slice, algo=None):
        if per_slice:
            return torch.einsum("tnk,tnf->tkf", A.double(), dY.double()).float()
        return torch.einsum("tnk,tnf->kf", A.double(), dY.double()).float()

    def act_fwd(self, x, act):
        return orc.ACTS[act](x)

    def act_bwd(self, x, dy, act):
        with torch.enable_grad():
            x = x.detach().clone().requires_grad_(True)
            orc.ACTS[act](x).backward(dy)
        return x.grad
