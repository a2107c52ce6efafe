"""CPU: the host sanitizer pass — `make -C tests/sanitize run`: the plain-C oracle rebuilt with
gcc -fsanitize=address,undefined under every golden fixture, and a sanitized C program that calls every entry point
of include/tmgcn.h with invalid / null / zero-size arguments (argument validation returns before any device work).
Never run on the GPU box (sanitizer runs are refused there); the kernels themselves are checked by the parity tests."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SAN = os.path.join(HERE, "sanitize")
BAD_MARKS = ("ERROR: AddressSanitizer", "runtime error:", "ERROR: LeakSanitizer", "SUMMARY: UndefinedBehaviorSanitizer")

pytestmark = pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("make") is None, reason="needs gcc and make")


def _make(target):
    p = subprocess.run(["make", "-C", SAN, target], capture_output=True, text=True, timeout=900)
    return p.returncode, p.stdout + p.stderr


def test_sanitizers_are_live(tmp_path):
    """Negative control: the same flags do report a heap overflow and a signed overflow."""
    src = tmp_path / "bad.c"
    src.write_text('#include <stdlib.h>\nint main(int c, char** v) { int* p = malloc(4 * sizeof(int)); int r = p[c + 3]; free(p); return r; }\n')
    exe = tmp_path / "bad"
    subprocess.check_call(["gcc", "-fsanitize=address,undefined", "-g", str(src), "-o", str(exe)])
    p = subprocess.run([str(exe)], capture_output=True, text=True, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0"})
    assert p.returncode != 0 and "AddressSanitizer" in p.stderr


def test_every_abi_entry_point_rejects_bad_arguments_under_asan_ubsan():
    rc, out = _make("run-abi")
    assert rc == 0, out[-3000:]
    assert "0 failures" in out and not any(m in out for m in BAD_MARKS), out[-3000:]


def test_c_oracle_under_asan_ubsan_passes_every_golden_fixture():
    rc, out = _make("run-oracle")
    assert rc == 0, out[-3000:]
    assert " passed" in out and not any(m in out for m in BAD_MARKS), out[-3000:]
