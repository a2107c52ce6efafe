#!/usr/bin/env python3
"""Where the time of the entry-major backward of the fused layers (l12_bwd_em_kernel) goes, block by block.

A development build of the library (-DTMGCN_L12_TRACE: build/variants/trace, made by
`make -C tm-gcn_amd/csrc OBJDIR=… LIB=… EXTRA=-DTMGCN_L12_TRACE`) leaves 100 MHz wall-clock stamps of the phases of every
block's row blocks in a device array; this script runs training steps of the chess data / a synthetic config on it, reads
the stamps of the LAST backward launch and prints the launch ramp, the phases' medians and the tail.  The traced library
must be the one the process loads: copy it over tm-gcn_amd/libtmgcn_hip.so on the (scratch) GPU box first.
    cp build/variants/trace/libtmgcn_hip.so tm-gcn_amd/ && python3 tools/l12_trace.py chess | S1
Stamps per block (thread 0): 0 entry; per row block k: 1+4k row pointers in LDS, 2+4k first (col, val) tile arrived,
3+4k first tile's gathered rows parked, 4+4k all tiles summed; 13 before the slab / ticket tail, 14 after it."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))


def model(which):
    import tmgcn_amd.layers as ehf
    if which == "chess":
        from _g10 import G10
        from tmgcn_amd import adjacency
        g = G10()
        k, i, j = g.raw
        Chat, _ = adjacency.build_adjacency(k, i, j, np.ones(len(k), np.float32), g.TT, g.N, M=None, window=10)
        A = adjacency.m_product_csr(Chat.slices(0, g.T), g.M)
        torch.manual_seed(0)
        m = ehf.EmbeddingGCN2(A, torch.from_numpy(g.X_train), torch.from_numpy(g.edges_train), torch.from_numpy(g.M),
                              hidden_feat=[6, 6, 3], condensed_W=True, use_Minv=False, nonlin2="selu")
        tgt = torch.from_numpy(g.target_train).cuda()
        crit = torch.nn.CrossEntropyLoss(weight=torch.from_numpy(g.class_weights).cuda())
        return m, crit, tgt
    import bench
    from tmgcn_amd import synth
    from tmgcn_amd.losses import WeightedCrossEntropy
    g = synth.dynamic_graph(**synth.CONFIGS[which], seed=0)
    spec = bench.EPOCH_MODELS[which]
    torch.manual_seed(0)
    m = ehf.EmbeddingGCN2(g.At_list(), torch.from_numpy(g.X), torch.from_numpy(g.edges), torch.from_numpy(g.M),
                          hidden_feat=spec["hidden"], nonlin2=spec["nonlin"], condensed_W=True, use_Minv=False)
    return m, WeightedCrossEntropy(torch.tensor([0.9, 0.1])).cuda(), torch.from_numpy(g.labels).cuda()


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "chess"
    fwd = "--fwd" in sys.argv[2:]                               # the entry-major FORWARD's stamps (0 entry, 1 row pointers, 2 (col, val), 3 parked, 4 summed, 5 before the stores, 14 end)
    from tmgcn_amd import _lib
    lib = _lib.load()
    fn = getattr(lib, "tmgcn_debug_l12_trace", None)
    if fn is None:
        raise SystemExit("this libtmgcn_hip.so was not built with -DTMGCN_L12_TRACE")
    fn.argtypes = [C.c_void_p, C.c_long, C.c_int]
    fn.restype = C.c_int
    m, crit, tgt = model(which)
    words = np.zeros(8192 * 16, np.uint64)
    for it in range(6):
        for p in m.parameters():
            p.grad = None
        if it == 5:
            torch.cuda.synchronize()
            fn(words.ctypes.data, 0, 1 | (2 if fwd else 0))   # clear
        m.loss(crit, tgt).backward()
    torch.cuda.synchronize()
    assert fn(words.ctypes.data, words.size, 2 if fwd else 0) == 0
    w = words.reshape(8192, 16).astype(np.int64)
    if fwd:
        w = w[w[:, 0] > 0]
        t0 = w[:, 0].min()
        q = lambda x: [round(float(v), 2) for v in np.percentile(x, [0, 10, 50, 90, 100])]
        d = lambda i, j: q((w[:, i] - w[:, j]) / 100.0)
        life = (w[:, 14] - w[:, 0]) / 100.0
        span = float((w[:, 14].max() - t0) / 100.0)
        rec = {"config": which, "kernel": "l12_fwd_em", "blocks": len(w), "unit": "us; [min, p10, median, p90, max]",
               "block_start_after_first": q((w[:, 0] - t0) / 100.0), "kernel_span_us": round(span, 2), "rowptr_in_lds": d(1, 0),
               "col_val_arrived": d(2, 1), "gathers_layer1_parked": d(3, 2), "first_tile_summed": d(4, 3), "further_tiles": d(5, 4),
               "stores_issued": d(14, 5), "block_lifetime": q(life), "block_lifetime_mean": round(float(life.mean()), 2),
               "resident_on_average": round(float(life.sum() / span), 1)}
        print(json.dumps(rec))
        with open(os.path.join(root, "gpurun_out", f"l12_trace_fwd_{which}.json"), "w") as f:
            json.dump(rec, f, indent=1)
        np.save(os.path.join(root, "gpurun_out", f"l12_trace_fwd_{which}.npy"), w - t0)
        return
    used = w[:, 0] > 0
    w = w[used]
    nb = len(w)
    t0 = w[:, 0].min()
    us = lambda x: (x - t0) / 100.0                          # 100 MHz -> us
    q = lambda x: [round(float(v), 2) for v in np.percentile(x, [0, 10, 50, 90, 100])]
    rec = {"config": which, "blocks": nb, "unit": "us; [min, p10, median, p90, max]",
           "block_start_after_first": q(us(w[:, 0])), "block_end_after_first": q(us(w[:, 14][w[:, 14] > 0])),
           "kernel_span_us": round(float(us(w[:, 14].max())), 2)}
    for k in range(3):
        have = w[:, 4 + 4 * k] > 0
        if not have.any():
            break
        prev = w[have, 0] if k == 0 else w[have, 4 * k]
        rec[f"row_block_{k}"] = {
            "blocks": int(have.sum()),
            "rowptr_in_lds": q((w[have, 1 + 4 * k] - prev) / 100.0),
            "col_val_arrived": q((w[have, 2 + 4 * k] - w[have, 1 + 4 * k]) / 100.0),
            "gathers_parked": q((w[have, 3 + 4 * k] - w[have, 2 + 4 * k]) / 100.0),
            "tiles_summed": q((w[have, 4 + 4 * k] - w[have, 3 + 4 * k]) / 100.0)}
    last = np.where(w[:, 4] > 0, w[:, 4], w[:, 0])
    for k in (1, 2):
        last = np.where(w[:, 4 + 4 * k] > 0, w[:, 4 + 4 * k], last)
    rec["row_math_until_tail"] = q((w[:, 13] - last) / 100.0)
    have = w[:, 14] > 0
    rec["tail_slab_and_tickets"] = q((w[have, 14] - w[have, 13]) / 100.0)
    rec["last_row_work_done_at"] = round(float(us(w[:, 13].max())), 2)
    life = (w[have, 14] - w[have, 0]) / 100.0
    rec["block_lifetime"] = q(life)
    rec["block_lifetime_mean"] = round(float(life.mean()), 2)
    rec["slot_time_sum_over_span"] = round(float(life.sum() / us(w[:, 14].max())), 1)      # blocks resident on average
    print(json.dumps(rec))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(root, "gpurun_out", f"l12_trace_{which}.npy"), w - t0)
    At = m.At.transpose()
    blk = At.row_block_runs() if At.row_blocks() is not None else None     # the list the backward was given (ops.layer12)
    if blk is not None:                                          # the partition's row blocks: rows, entries, longest row, rows over 64
        rp = At.rowptr.cpu().numpy()
        b = blk.cpu().numpy()                                    # (first row, rows), in the order the kernel hands them out
        ln = np.diff(rp)
        stats = np.array([[n, rp[f + n] - rp[f], ln[f:f + n].max(initial=0), (ln[f:f + n] > 64).sum()] for f, n in b])
        np.save(os.path.join(root, "gpurun_out", f"l12_trace_{which}_row_blocks.npy"), stats)
    with open(os.path.join(root, "gpurun_out", f"l12_trace_{which}.json"), "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
