#!/usr/bin/env python3
"""Interleaved A/B timing of the edge-head kernels (P4: forward, dZ, dU) across library variants
(tools/ab_variants.sh), at the Reddit-LP-shaped size (R = 65 x 3800 rows, F = 6, C = 2, E = 3.25 M
labelled edges) unless overridden, with a bitwise / tolerance comparison between the variants.
   python tools/ab_edge_head.py [variant ...]          env: AB_F, AB_C, AB_E, AB_T, AB_N"""
import ctypes as C
import glob
import os
import statistics
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p, i32, i64 = C.c_void_p, C.c_int32, C.c_int64


def load(path):
    lib = C.CDLL(path)
    lib.tmgcn_edge_head_fwd_f32.argtypes = [p, p, p, p, p, i64, i32, i32, p]
    lib.tmgcn_edge_head_bwd_workspace_bytes.restype = i64
    lib.tmgcn_edge_head_bwd_workspace_bytes.argtypes = [i64, i32, i32]
    lib.tmgcn_edge_head_bwd_f32.argtypes = [p, p, p, p, p, p, p, p, p, i64, i64, i32, i32, p, i64, p]
    if hasattr(lib, "tmgcn_edge_head_fwd_i32_f32"):
        lib.tmgcn_edge_head_fwd_i32_f32.argtypes = lib.tmgcn_edge_head_fwd_f32.argtypes
        lib.tmgcn_edge_head_bwd_i32_f32.argtypes = lib.tmgcn_edge_head_bwd_f32.argtypes
    return lib


names = sys.argv[1:] or sorted(os.path.basename(os.path.dirname(f)) for f in glob.glob(root + "/build/variants/*/libtmgcn_hip.so"))
libs = {"default": load(root + "/tm-gcn_amd/libtmgcn_hip.so")}
for n in names:
    libs[n] = load(f"{root}/build/variants/{n}/libtmgcn_hip.so")
libs["i32"] = libs["default"]      # the same library through its 32-bit-index entry points

F, Cn = int(os.environ.get("AB_F", 6)), int(os.environ.get("AB_C", 2))
T, N = int(os.environ.get("AB_T", 65)), int(os.environ.get("AB_N", 3800))
E = int(os.environ.get("AB_E", 3_249_165))
R = T * N
g = torch.Generator(device="cuda").manual_seed(0)
t = torch.randint(0, T, (E,), device="cuda", generator=g)
src = t * N + torch.randint(0, N, (E,), device="cuda", generator=g)
dst = t * N + torch.randint(0, N, (E,), device="cuda", generator=g)
Z = torch.randn(R, F, device="cuda", generator=g)
U = torch.randn(2 * F, Cn, device="cuda", generator=g)
dout = torch.randn(E, Cn, device="cuda", generator=g)
# inverted index: entries 2e (row = src[e]) and 2e+1 (row = dst[e]) grouped by row
rows = torch.stack([src, dst], 1).reshape(-1)
order = torch.argsort(rows, stable=True)
eidx = order.contiguous()
eptr = torch.zeros(R + 1, dtype=torch.int64, device="cuda")
eptr[1:] = torch.cumsum(torch.bincount(rows, minlength=R), 0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda x: C.c_void_p(x.data_ptr())
ws = torch.empty(max(int(libs["default"].tmgcn_edge_head_bwd_workspace_bytes(E, F, Cn)), 1) * 2, dtype=torch.uint8, device="cuda")


src32, dst32, eptr32, eidx32 = src.int(), dst.int(), eptr.int(), eidx.int()


def run(lib, which, out, name=""):
    if name == "i32":
        if which == "fwd":
            return lib.tmgcn_edge_head_fwd_i32_f32(ptr(Z), ptr(src32), ptr(dst32), ptr(U), ptr(out), E, F, Cn, st)
        dz, du = (ptr(out), None) if which == "dZ" else (None, ptr(out))
        return lib.tmgcn_edge_head_bwd_i32_f32(ptr(Z), ptr(src32), ptr(dst32), ptr(U), ptr(dout), ptr(eptr32), ptr(eidx32),
                                               dz, du, R, E, F, Cn, ptr(ws), ws.numel(), st)
    if which == "fwd":
        return lib.tmgcn_edge_head_fwd_f32(ptr(Z), ptr(src), ptr(dst), ptr(U), ptr(out), E, F, Cn, st)
    if which == "dZ":
        return lib.tmgcn_edge_head_bwd_f32(ptr(Z), ptr(src), ptr(dst), ptr(U), ptr(dout), ptr(eptr), ptr(eidx), ptr(out),
                                           None, R, E, F, Cn, ptr(ws), ws.numel(), st)
    return lib.tmgcn_edge_head_bwd_f32(ptr(Z), ptr(src), ptr(dst), ptr(U), ptr(dout), ptr(eptr), ptr(eidx), None,
                                       ptr(out), R, E, F, Cn, ptr(ws), ws.numel(), st)


shapes = {"fwd": (E, Cn), "dZ": (R, F), "dU": (2 * F, Cn)}
res, outs = {}, {}
for which in ("fwd", "dZ", "dU"):
    for name, lib in libs.items():
        outs[(which, name)] = torch.zeros(*shapes[which], device="cuda")
        assert run(lib, which, outs[(which, name)], name) == 0, (which, name)
    torch.cuda.synchronize()
    for rnd in range(9):
        for name, lib in libs.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            run(lib, which, outs[(which, name)], name)
            e.record()
            torch.cuda.synchronize()
            res.setdefault((which, name), []).append(s.elapsed_time(e) * 1e3)
cat = torch.cat([Z[src], Z[dst]], 1).double()
truth = {"fwd": cat @ U.double(), "dU": cat.t() @ dout.double()}
dcat = dout.double() @ U.double().t()
dz = torch.zeros(R, F, dtype=torch.float64, device="cuda")
dz.index_add_(0, src, dcat[:, :F])
dz.index_add_(0, dst, dcat[:, F:])
truth["dZ"] = dz
print(f"R={R} F={F} C={Cn} E={E}")
for (which, name), us in res.items():
    o = outs[(which, name)]
    err = float((o.double() - truth[which]).abs().max() / truth[which].abs().max())
    same = "" if name == "default" else f"  bitwise-equal-to-default={bool(torch.equal(o, outs[(which, 'default')]))}"
    print(f"{which:4s} {name:12s} median {statistics.median(us):8.1f} us   min {min(us):8.1f} us   err vs fp64 {err:.1e}{same}")
