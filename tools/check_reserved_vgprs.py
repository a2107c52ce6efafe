#!/usr/bin/env python3
"""Build-time check of the asynchronous staging scheme (tm-gcn_amd/csrc/async_stage.h).

The stream kernels (gemm_bf16x3, mtransform_bf16x3) park in-flight global loads in FIXED registers at
the top of the register file (v192..v255) from inline asm.  That is sound only if no
COMPILER-generated instruction of those kernels names a register of that zone, and if the kernels
call no out-of-line code (which would have a register allocation of its own).  hipcc offers no way
to enforce this, so it is checked on what the build produced:

  --asm FILE.s …   the device assembly hipcc emitted for the object that ships (`-save-temps=obj`
                   by-product of the very compile that produced gemm.o / mtransform.o: the Makefile
                   runs this form and FAILS THE BUILD on a violation).  Inline-asm blocks are
                   delimited by ;ASMSTART / ;ASMEND there, so "compiler-generated" is exact.
  --lib LIB.so     the shipped library: its gfx950 code objects are extracted and disassembled
                   (llvm-objdump).  A disassembly has no asm markers, so this form allows exactly the
                   three instruction shapes of async_stage.h to touch the zone (global_load_dwordx4
                   INTO a reserved quad; v_mov_b32 / v_mul_f32 reading ONE reserved register into an
                   ordinary one) and flags everything else.  The Makefile runs it after linking.
  (no arguments)   compile gemm.hip / mtransform.hip to assembly with the flags the Makefile would use
                   (`make print-hipcc`), run the --asm check, then --lib on the in-tree library.
  --selftest       negative tests of the checker itself (violations must be reported).

Every kernel is scanned from its label to .Lfunc_end (not to the first s_endpgm: early-exit blocks
are followed by more code).  Kernels not listed below whose inline asm names v192+ fail the check
too (a new user of async_stage.h must be listed).  Exit status 0 = clean, 1 = violation.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tm-gcn_amd", "csrc")
LLVM = os.environ.get("TMGCN_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ZONE = 192                      # lowest register async_stage.h ever names
KERNELS = {                     # source file -> {mangled-name fragment: (first reserved VGPR, expected instantiations)}
    "gemm.hip": {"gemm_bf16x3_kernel": (192, 4)},
    "mtransform.hip": {"mtransform_bf16x3_kernel": (192, 1)},
}
# Calls: s_swappc_b64 / s_call_b64, and in compiler assembly any pc-relative reference to a SYMBOL
# (@rel32@ / @gotpcrel32@: the address of a function is being formed, e.g. for a tail call).
# s_setpc_b64 alone is NOT a call: large kernels use s_getpc/s_add/s_setpc for long branches to
# their own .LBB labels.
CALLS = re.compile(r"\b(s_swappc_b64|s_call_b64)\b|@(?:got)?(?:pc)?rel32@")


def regs_of(text):
    """VGPR indices named in an instruction's operand text (v7, v[4:7]; not a7 / s7 / acc)."""
    out = [int(x) for x in re.findall(r"(?<![\w.])v(\d+)\b", text)]
    for a, b in re.findall(r"(?<![\w.])v\[(\d+):(\d+)\]", text):
        out += list(range(int(a), int(b) + 1))
    return out


# ------------------------------------------------------------------------------------- --asm
def functions_of_asm(text):
    """(name, body) of every function of a compiler-emitted .s: label .. .Lfunc_endN."""
    for m in re.finditer(r"^([A-Za-z_]\w*):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        yield m.group(1), m.group(2)


def scan_asm_body(body, first):
    """(highest VGPR named by compiler code, violations, registers named inside inline asm >= ZONE)."""
    inasm, top, hits, asm_zone = False, -1, [], set()
    for line in body.split("\n"):
        code = line.split(";")[0] if "ASMSTART" not in line and "ASMEND" not in line else line
        if "ASMSTART" in line:
            inasm = True
            continue
        if "ASMEND" in line:
            inasm = False
            continue
        ins = code.strip()
        if not ins or ins.startswith(".") or ins.endswith(":"):
            continue
        if inasm:
            asm_zone.update(r for r in regs_of(ins) if r >= ZONE)
            continue
        if CALLS.search(ins):
            hits.append("call to out-of-line code: " + ins)
        regs = regs_of(ins)
        if regs:
            top = max(top, max(regs))
            if first is not None and max(regs) >= first:
                hits.append(ins)
    return top, hits, asm_zone


def check_asm(paths, kernels, quiet=False):
    bad, seen = 0, {}
    for path in paths:
        text = open(path).read()
        for name, body in functions_of_asm(text):
            frag = next((f for f in kernels if f in name), None)
            first = kernels[frag][0] if frag else None
            top, hits, asm_zone = scan_asm_body(body, first)
            if frag is None:
                if asm_zone:
                    print(f"{name}: inline asm names v{min(asm_zone)}.. but the kernel is not listed in "
                          f"tools/check_reserved_vgprs.py: VIOLATION")
                    bad += 1
                continue
            seen[frag] = seen.get(frag, 0) + 1
            if not asm_zone:
                hits.append("no inline-asm use of the reserved zone found (is this still an async_stage.h kernel?)")
            if not quiet or hits:
                print(f"{name}: compiler code uses v0..v{top}, reserved zone starts at v{first}: "
                      f"{'OK' if not hits else f'{len(hits)} VIOLATIONS, e.g. ' + hits[0]}")
            bad += bool(hits)
    for frag, (_first, expect) in kernels.items():
        if seen.get(frag, 0) == 0:
            print(f"kernel {frag}: not found in {', '.join(os.path.basename(p) for p in paths)}: VIOLATION")
            bad += 1
        elif expect and seen[frag] != expect:
            print(f"kernel {frag}: {seen[frag]} instantiations, expected {expect} (update KERNELS): VIOLATION")
            bad += 1
    return bad


# ------------------------------------------------------------------------------------- --lib
ALLOWED = (
    # async_stage.h TMGCN_Q_LOAD / TMGCN_Q_LOAD_S: 16-byte load INTO a reserved quad
    (re.compile(r"^global_load_dwordx4\s+v\[(\d+):(\d+)\],\s*(.*)$"), "load"),
    # TMGCN_Q_READ: v_mov_b32 ordinary, reserved
    (re.compile(r"^v_mov_b32(?:_e32)?\s+v(\d+),\s*v(\d+)\s*$"), "mov"),
    # TMGCN_Q_READ_MUL: v_mul_f32 ordinary, reserved, ordinary
    (re.compile(r"^v_mul_f32(?:_e32)?\s+v(\d+),\s*v(\d+),\s*v(\d+)\s*$"), "mul"),
)


def allowed_in_disassembly(ins, first):
    for rx, kind in ALLOWED:
        m = rx.match(ins)
        if not m:
            continue
        if kind == "load":
            a, b = int(m.group(1)), int(m.group(2))
            return a >= first and b == a + 3 and (a - first) % 4 == 0 and all(r < first for r in regs_of(m.group(3)))
        if kind == "mov":
            return int(m.group(1)) < first <= int(m.group(2))
        if kind == "mul":
            return int(m.group(1)) < first <= int(m.group(2)) and int(m.group(3)) < first
    return False


def scan_disassembly_body(lines, first):
    hits, zone_used = [], False
    for line in lines:
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        if CALLS.search(ins):
            hits.append("call to out-of-line code: " + ins)
        regs = regs_of(ins)
        if regs and max(regs) >= first:
            zone_used = True
            if not allowed_in_disassembly(ins, first):
                hits.append(ins)
    return hits, zone_used


def check_lib(lib, kernels, quiet=False, allow_missing=False):
    objdump = os.path.join(LLVM, "llvm-objdump")
    bad, seen = 0, {}
    with tempfile.TemporaryDirectory() as d:
        copy = os.path.join(d, os.path.basename(lib))
        shutil.copy(lib, copy)                       # --offloading extracts next to its input
        subprocess.check_call([objdump, "--offloading", copy], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(f for f in os.listdir(d) if "amdgcn" in f)
        if not objs:
            print(f"{lib}: no gfx950 code object found: VIOLATION")
            return 1
        for o in objs:
            dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", os.path.join(d, o)], capture_output=True, text=True,
                                 check=True).stdout
            cur, body = None, []

            def flush():
                nonlocal bad
                if cur is None:
                    return
                frag = next((f for f in kernels if f in cur), None)
                if frag is None:
                    return
                first = kernels[frag][0]
                hits, zone_used = scan_disassembly_body(body, first)
                if not zone_used:
                    hits.append("the reserved zone is never used (is this still an async_stage.h kernel?)")
                seen[frag] = seen.get(frag, 0) + 1
                if not quiet or hits:
                    print(f"{os.path.basename(lib)}:{cur}: only async_stage.h's own instruction shapes touch v{first}..: "
                          f"{'OK' if not hits else f'{len(hits)} VIOLATIONS, e.g. ' + hits[0]}")
                bad += bool(hits)

            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:\s*$", line)
                if m:
                    flush()
                    cur, body = m.group(1), []
                elif cur is not None:
                    body.append(line)
            flush()
    for frag, (_first, expect) in kernels.items():
        if seen.get(frag, 0) == 0:
            if allow_missing:       # a tuning-variant library built from a subset of the sources
                print(f"kernel {frag}: not in {lib} (allowed: --allow-missing)")
                continue
            print(f"kernel {frag}: not found in {lib}: VIOLATION")
            bad += 1
        elif expect and seen[frag] != expect:
            print(f"kernel {frag}: {seen[frag]} instantiations in {lib}, expected {expect}: VIOLATION")
            bad += 1
    return bad


# ------------------------------------------------------------------------------------- default / selftest
def makefile_compile_command():
    """`$(HIPCC) $(CXXFLAGS)` exactly as tm-gcn_amd/csrc/Makefile expands them (ARCH / EXTRA / HIPCC
    overrides from the environment or the make command line included)."""
    out = subprocess.run(["make", "-s", "--no-print-directory", "-C", CSRC, "print-hipcc"], capture_output=True, text=True,
                         check=True).stdout.strip()
    return out.split()


def all_kernels():
    k = {}
    for v in KERNELS.values():
        k.update(v)
    return k


def default_run():
    cmd = makefile_compile_command()
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for src, kernels in KERNELS.items():
            out = os.path.join(d, src.replace(".hip", ".s"))
            subprocess.check_call(cmd + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
            bad += check_asm([out], kernels)
    lib = os.path.join(ROOT, "tm-gcn_amd", "libtmgcn_hip.so")
    if os.path.exists(lib):
        bad += check_lib(lib, all_kernels())
    return bad


GOOD_ASM = """\t.text
_Z9k_exampleILi1EEvPf:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\tv_mov_b32_e32 v1, 0
\t;;#ASMSTART
\tglobal_load_dwordx4 v[192:195], v[2:3], off
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_2
\ts_endpgm
.LBB0_2:
\t;;#ASMSTART
\tv_mov_b32 v4, v192
\t;;#ASMEND
\tv_add_f32_e32 v5, v4, v1
\ts_endpgm
.Lfunc_end0:
"""


def selftest():
    k = {"k_example": (192, 1)}
    fails = []

    def expect(name, text, want_bad, kernels=k):
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "t.s")
            open(p, "w").write(text)
            bad = check_asm([p], kernels, quiet=True)
        if bool(bad) != want_bad:
            fails.append(name)
        print(f"selftest {name}: {'flagged' if bad else 'clean'} ({'as expected' if bool(bad) == want_bad else 'WRONG'})")

    expect("clean kernel", GOOD_ASM, False)
    expect("compiler instruction names a reserved register",
           GOOD_ASM.replace("v_mov_b32_e32 v1, 0", "v_mov_b32_e32 v200, 0"), True)
    expect("violation behind the first s_endpgm (early-exit block)",
           GOOD_ASM.replace("v_add_f32_e32 v5, v4, v1", "v_add_f32_e32 v5, v193, v1"), True)
    expect("register range reaching into the zone",
           GOOD_ASM.replace("v_mov_b32_e32 v1, 0", "ds_read_b128 v[190:193], v1"), True)
    expect("call to out-of-line code",
           GOOD_ASM.replace("v_mov_b32_e32 v1, 0", "s_swappc_b64 s[30:31], s[16:17]"), True)
    expect("listed kernel missing from the file", GOOD_ASM, True, {"k_other": (192, 1)})
    expect("unlisted kernel using the zone from inline asm", GOOD_ASM, True, {})
    expect("wrong number of instantiations", GOOD_ASM, True, {"k_example": (192, 2)})
    # the disassembly whitelist
    for ins, ok in (("global_load_dwordx4 v[192:195], v[2:3], off", True),
                    ("global_load_dwordx4 v[196:199], v2, s[4:5]", True),
                    ("global_load_dwordx4 v[194:197], v[2:3], off", False),       # not a quad of the scheme
                    ("global_load_dwordx4 v[192:195], v[200:201], off", False),   # address from the zone
                    ("v_mov_b32_e32 v7, v201", True), ("v_mov_b32_e32 v201, v7", False),
                    ("v_mul_f32_e32 v7, v201, v3", True), ("v_mul_f32_e32 v7, v3, v201", False),
                    ("v_fma_f32 v7, v201, v3, v4", False), ("ds_read_b128 v[192:195], v1", False)):
        got = allowed_in_disassembly(ins, 192)
        if got != ok:
            fails.append("whitelist: " + ins)
        print(f"selftest whitelist {ins!r}: {'allowed' if got else 'flagged'} ({'as expected' if got == ok else 'WRONG'})")
    return len(fails)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--asm", nargs="+", help="compiler-emitted device assembly file(s) to check")
    ap.add_argument("--lib", help="built shared library whose gfx950 code objects are disassembled and checked")
    ap.add_argument("--kernels", nargs="+", metavar="FRAGMENT:FIRST[:COUNT]",
                    help="override the kernel table (name fragment, first reserved VGPR, expected instantiations; 0 = any)")
    ap.add_argument("--allow-missing", action="store_true",
                    help="--lib: a listed kernel may be absent (variant libraries built from a subset of the sources)")
    ap.add_argument("--selftest", action="store_true")
    a = ap.parse_args()
    if a.selftest:
        return 1 if selftest() else 0
    kernels = all_kernels()
    if a.kernels:
        kernels = {}
        for spec in a.kernels:
            parts = spec.split(":")
            kernels[parts[0]] = (int(parts[1]), int(parts[2]) if len(parts) > 2 else 0)
    bad = 0
    if a.asm:
        if not a.kernels:  # only the kernels whose source the given files come from
            names = {os.path.basename(p).split("-")[0].split(".")[0] + ".hip" for p in a.asm}
            picked = {}
            for n in names:
                picked.update(KERNELS.get(n, {}))
            kernels = picked or kernels
        bad += check_asm(a.asm, kernels)
    if a.lib:
        bad += check_lib(a.lib, kernels, allow_missing=a.allow_missing)
    if not a.asm and not a.lib:
        bad += default_run()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
