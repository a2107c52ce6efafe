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

    def spmm(self, A, X, tag=None):
        return orc.slice_spmm(A.to_coo_list(), X.double())

    def gemm(self, A, W, trans_w=False, act=None, want_pre=False):
        Wd = W.double().transpose(-1, -2) if trans_w else W.double()
        pre = torch.matmul(A.double(), Wd).float()
        Y = orc.ACTS[act](pre) if act else pre
        return (Y, pre if act else None) if want_pre else Y

    def gemm_dw(self, A, dY, per_slice):
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
