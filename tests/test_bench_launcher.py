"""bench.py's own multi-rank launcher and its deadline, exercised without a GPU:
`python bench.py --gpus N` with no torchrun around it must start N fresh ranks itself, relay their
failure as a non-zero exit code, and a stalled job must end — stacks dumped, non-zero — inside the
deadline instead of hanging until somebody's timeout ("What's missing" #1)."""
import os
import subprocess
import sys
import time

from _util import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_self_launch_starts_the_ranks_and_relays_their_exit_code():
    """No GPU here: every rank stops at `bench.py needs MI355X GPUs`; the parent (which never imports
    torch) must have launched 2 ranks through torch.distributed.run and must come back non-zero."""
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--deadline", "120", "--watchdog", "0"], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "launching 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    assert "[bench r0" in r.stderr and "[bench r1" in r.stderr            # both ranks started
    assert "needs MI355X GPUs" in r.stderr
    assert r.stdout.strip() == ""                                          # no JSON line from a failed job
    assert time.time() - t0 < 200


def test_a_stalled_job_ends_non_zero_inside_the_deadline():
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--deadline", "8", "--watchdog", "3", "--selftest-stall"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert r.returncode != 0
    assert "selftest: stalling on purpose" in r.stderr
    assert "watchdog: still running after 3 s" in r.stderr                 # soft dump first
    assert "Timeout (0:00:08)" in r.stderr                                 # faulthandler's hard deadline fired, with stacks
    assert took < 280, took                                                # generous: a cold `import torch` in the launcher alone can take 1-2 min


def test_single_rank_form_rejects_a_world_size_mismatch():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_traffic_leg_reduces_the_counter_passes(tmp_path, monkeypatch):
    """bench.measure_traffic: two child passes under `rocprofv3 --pmc …`, reduced to fabric-side bytes
    per FORWARD launch of the dominant kernel (FETCH_SIZE x 2 + WRITE_SIZE, KiB).  Here a stand-in
    `rocprofv3` on PATH writes the counter CSVs a real pass produces (values of profiles/pmc_traffic.json),
    so the pairing of forward / backward dispatches, the unit handling and the in-pass calibration
    are checked without a GPU; a failing pass must come back as (None, reason)."""
    import stat
    import bench
    fake = tmp_path / "bin"
    fake.mkdir()
    script = fake / "rocprofv3"
    script.write_text('''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
counter, out = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
if os.environ.get("FAKE_ROCPROF_FAIL"):
    sys.exit(7)
os.makedirs(os.path.join(out, "host", "1"), exist_ok=True)
rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value,Grid_Size"]
fused = "void tmgcn::spmm_gemm_kernel<32, 4, 16>(tmgcn::FusedArgs)"
band = "void tmgcn::mtransform_band_kernel<16, 4, 4, false>(tmgcn::MtArgs)"
vals = {"FETCH_SIZE": {fused: [270883771.9, 271501397.5] * 3, band: [8001143.5, 8112556.4] * 3},
        "WRITE_SIZE": {fused: [32031017.0, 16031017.2] * 3, band: [16000000.2] * 6}}[counter]
i = 0
for step in range(3):
    for name in (band, fused, fused, band):
        i += 1
        rows.append(f'{i},"{name}",{counter},{vals[name].pop(0)},1024')
open(os.path.join(out, "host", "1", "x_counter_collection.csv"), "w").write("\\n".join(rows) + "\\n")
''')
    script.chmod(script.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(fake) + os.pathsep + os.environ["PATH"])
    args = bench.parse([])
    src, why = bench.measure_traffic(args)
    assert why is None
    assert src["forward_bytes"] == int(270883771.9 * 1024 * 2 + 32031017.0 * 1024)    # the forward launch (larger WRITE_SIZE)
    assert src["kind"] == "measured in this run" and src["dispatches"] == 6
    assert abs(src["fetch_x2_calibration_on_band_mtransform"] - 2.0) < 1e-3        # 16.384 GB read exactly / raw counter
    assert src["backward_bytes"] == int(271501397.5 * 1024 * 2 + 16031017.2 * 1024)
    # another workload's passes (a side leg): the child command line carries ITS shape, the calibration its slab
    src, why = bench.measure_traffic(args, what="T128", nodes=250_000, slices_per_gpu=128)
    assert why is None and abs(src["fetch_x2_calibration_on_band_mtransform"] - 2.0) < 1e-3
    monkeypatch.setenv("FAKE_ROCPROF_FAIL", "1")
    src, why = bench.measure_traffic(args)
    assert src is None and "exited 7" in why


def test_launch_roofline_takes_measured_bytes_only_below_0p8_of_the_model():
    """`if FETCH_SIZE + WRITE_SIZE is < 0.8x the gather model, use the measured bytes`; a launch is never quoted
    above what it moved (the skewed leg's backward stood at 1.11x peak on model bytes)."""
    import bench
    model = 557.4e9
    a = bench.launch_roofline(model, 0.97 * model, 80.0)
    assert a["basis"] == "model" and abs(a["frac"] - model / 0.080 / 8e12) < 1e-9
    b = bench.launch_roofline(model, 300e9, 62.68)                            # hubs' rows re-read from cache
    assert b["basis"] == "measured" and b["frac"] < 0.61 and b["frac_model"] > 1.1
    c = bench.launch_roofline(model, None, 62.68)                             # nothing measured: the model, flagged by frac_model == frac
    assert c["basis"] == "model" and c["measured_bytes"] is None
    d = bench.launch_roofline(model, 0.87 * model, 62.68)                     # 0.87x: not below 0.8x, but the model says 1.11x peak
    assert d["basis"] == "measured" and d["frac"] < 1.0 < d["frac_model"]


def test_side_leg_record_is_composed_from_the_child_line(monkeypatch):
    """bench.measure_leg without a GPU: the child run and its two PMC passes are replaced by canned records; the leg's
    frac is (bytes used forward + backward) / (their time) / peak with each launch on its prescribed basis."""
    import bench
    args = bench.parse([])
    child = {"config": {"workload": "w", "row_lengths": {"mean": 4.0}}, "kernels_ms": {"mtransform": 5.9},
             "roofline": {"bytes_per_edge_slice": 647.0, "edge_slices_per_launch": 1.3e8, "forward_launch_ms": 20.0,
                          "backward_launch_ms": 18.0}, "ms_per_step": 60.0, "steps": 5, "value": 1e9,
             "verify": {"ok": True, "max_rel_err_Y": 1e-7, "max_rel_err_dX": 1e-7, "max_rel_err_dW": 1e-6, "seconds": 3.0}}
    seen = {}

    def fake_child(flags, what, timeout=420):
        seen["flags"] = flags
        return child, 0, None

    monkeypatch.setattr(bench, "run_child", fake_child)
    monkeypatch.setattr(bench, "measure_traffic", lambda a, what="x", **over: ({"forward_bytes": 50e9, "backward_bytes": 80e9}, None))
    rec = bench.measure_leg(args, "real_structure")
    assert "chess_tiled" in seen["flags"] and rec["verify_ok"] is True
    model = 647.0 * 1.3e8
    assert rec["forward"]["basis"] == "measured" and rec["backward"]["basis"] == "model"      # 50 < 0.8 x 84.1 <= 80
    assert abs(rec["frac"] - (50e9 + model) / 0.038 / 8e12) < 1e-9
    assert abs(rec["mfma"]["flops_per_launch"] - 2 * (1.3e8 / 4.0) * 128 * 128) < 1
    rec = bench.measure_leg(args, "T128")
    assert seen["flags"][seen["flags"].index("--slices-per-gpu") + 1] == "128" and seen["flags"][seen["flags"].index("--nodes") + 1] == "250000"


def test_timeline_gaps_tool_unions_overlapping_kernels(tmp_path):
    """tools/timeline_gaps.py on a synthetic kernel trace: two steps, kernels of two streams overlapping,
    one idle gap — busy time is the UNION of the intervals, the gap is attributed to its neighbours."""
    import json
    import subprocess
    import sys
    rows = ['"Kind","Stream_Id","Kernel_Name","Start_Timestamp","End_Timestamp"']
    t = 1_000_000

    def k(name, start_us, dur_us, stream=0):
        rows.append(f'"KERNEL_DISPATCH",{stream},"{name}",{t + start_us * 1000},{t + (start_us + dur_us) * 1000}')

    for step in range(3):
        o = step * 10_000
        k("void tmgcn::mtransform_band_kernel<16, 4, 4, false>(tmgcn::MtArgs)", o + 0, 1000)
        k("void tmgcn::spmm_gemm_kernel<32, 4, 16>(tmgcn::FusedArgs)", o + 1000, 3000)
        k("rcclGenericKernel(x)", o + 2000, 1500, stream=1)                       # fully inside the previous kernel
        k("void tmgcn::spmm_gemm_kernel<32, 4, 16>(tmgcn::FusedArgs)", o + 3800, 3000, stream=2)   # overlaps its tail
        k("void tmgcn::mtransform_band_kernel<16, 4, 4, false>(tmgcn::MtArgs)", o + 7300, 1000)    # after a 500 us gap
        k("tmgcn::gemm_dw_bf16x3_kernel(tmgcn::DwArgs)", o + 8300, 1700)
    p = tmp_path / "x_kernel_trace.csv"
    p.write_text("\n".join(rows) + "\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timeline_gaps.py"), str(p), "--steps", "2"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    steps = json.loads(r.stdout)
    assert len(steps) == 2
    s = steps[-1]
    assert s["wall_ms"] == 10.0 and s["busy_union_ms"] == 9.5 and s["idle_ms"] == 0.5
    assert s["kernel_ms_sum"] == 11.2                                             # 1 + 3 + 1.5 + 3 + 1 + 1.7: overlaps counted twice
    assert s["largest_gaps_ms"][0] == {"ms": 0.5, "after": "spmm_gemm_kernel", "before": "mtransform_band_kernel"}
    assert s["by_kernel"]["spmm_gemm_kernel"] == {"launches": 2, "ms": 6.0}
