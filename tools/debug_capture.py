"""Does a captured training step survive an eager step in front of it whose loss / output the caller still holds?
(Round 5: it crashed in hipStreamEndCapture — stale AccumulateGrad nodes bound to the eager stream; graphs.py now uses
autograd.grad inside the capture.)  Each variant runs in a child process.  python3 tools/debug_capture.py"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.abspath(__file__)) if "__file__" in globals() else os.getcwd()
CHILD = r'''
import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np, torch
from _g10 import G10
from _util import golden
import tmgcn_amd.layers as ehf
from tmgcn_amd import adjacency
from tmgcn_amd.graphs import GraphedTrainStep
from tmgcn_amd.optim import FusedSGD
variant = sys.argv[1]
g = G10(); d = golden("g11_chess_train300")
k, i, j = g.raw
Chat, _ = adjacency.build_adjacency(k, i, j, np.ones(len(k), np.float32), g.TT, g.N, M=None, window=10)
A_train = adjacency.m_product_csr(Chat.slices(0, g.T), g.M)
torch.manual_seed(71)
m = ehf.EmbeddingGCN2(A_train, torch.from_numpy(g.X_train), torch.from_numpy(g.edges_train), torch.from_numpy(g.M), hidden_feat=[6, 6, 3], condensed_W=True, use_Minv=False, nonlin2="selu")
tgt = torch.from_numpy(g.target_train).cuda()
crit = torch.nn.CrossEntropyLoss(weight=torch.from_numpy(g.class_weights).cuda())
opt = (torch.optim.SGD if "torchsgd" in variant else FusedSGD)(m.parameters(), lr=0.01, momentum=0.9)
opt.zero_grad(set_to_none=True)
if "unit" in variant:
    from tmgcn_amd import ops
    loss, out = m.loss(crit, tgt, want_logits=True, unit_grad=True)
    loss.backward(gradient=ops.unit_gradient(tgt.device))
else:
    loss, out = m.loss(crit, tgt, want_logits=True)
    loss.backward()
opt.step()
torch.cuda.synchronize()
print(variant, "eager ok", float(loss), flush=True)
w = 0 if "w0" in variant else 1
if "del" in variant:
    del loss, out
step = GraphedTrainStep(m, crit, opt, tgt, warmup=w, keep_logits="nologits" not in variant)
print(variant, "captured", flush=True)
for _ in range(3): l = step()
torch.cuda.synchronize()
print(variant, "replayed ok", float(l), flush=True)
'''
for v in ["w1", "w0", "w0_unit", "w0_torchsgd", "w0_del"]:
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": os.getcwd()}, v], capture_output=True, text=True, timeout=600)
    print("=====", v, "rc", r.returncode)
    print((r.stdout or "")[-600:])
    err = (r.stderr or "")
    print("\n".join(l for l in err.splitlines() if "amdgpu.ids" not in l and "Extension modules" not in l)[-700:])
