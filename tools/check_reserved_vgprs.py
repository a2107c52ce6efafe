#!/usr/bin/env python3
"""Build-time check of the asynchronous staging scheme (csrc/async_stage.h): in every kernel that parks
in-flight global loads in the reserved top of the register file (v192.. / v224..), no
COMPILER-generated instruction may name a register of that zone — it receives data asynchronously.  Compiles the kernels to assembly and
scans everything outside inline-asm blocks.   python tools/check_reserved_vgprs.py  -> exit 0 / 1"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tm-gcn_amd", "csrc")
KERNELS = {  # source file -> {mangled-name fragment: first reserved VGPR}
    "gemm.hip": {"gemm_bf16x3_kernel": 192},
    "mtransform.hip": {"mtransform_bf16x3_kernel": 192},
}
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-mllvm", "-pragma-unroll-threshold=200000", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only"]


def scan(body, first):
    """(highest VGPR named, offending lines) of the compiler-generated instructions of one kernel body."""
    inasm, top, hits = False, 0, []
    for line in body.split("\n"):
        if "ASMSTART" in line:
            inasm = True
        elif "ASMEND" in line:
            inasm = False
        elif not inasm:
            regs = [int(x) for x in re.findall(r"\bv(\d+)\b", line)]
            for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", line):
                regs += [int(a), int(b)]
            if regs:
                top = max(top, max(regs))
                if max(regs) >= first:
                    hits.append(line.strip())
    return top, hits


def main():
    bad = 0
    for src, kernels in KERNELS.items():
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.check_call(["hipcc", *FLAGS, os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
            text = open(out).read()
        for frag, first in kernels.items():
            found = list(re.finditer(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)s_endpgm" % frag, text, re.S | re.M))
            if not found:
                print(f"{src}: kernel {frag} not found")
                bad += 1
            for m in found:  # every instantiation of a templated kernel
                top, hits = scan(m.group(2), first)
                print(f"{m.group(1)}: compiler code uses v0..v{top}, reserved zone starts at v{first}: "
                      f"{'OK' if not hits else f'{len(hits)} VIOLATIONS, e.g. ' + hits[0]}")
                bad += bool(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
