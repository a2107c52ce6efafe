"""The C-ABI used the way a non-Python host would use it: tests/abi_driver/driver.c (plain C99 +
the HIP runtime, no torch) is compiled against include/tmgcn.h, linked with libtmgcn_hip.so and —
as the checker — the C oracle, and run on the MI355X.  It drives one layer forward and backward
(M-transform, batched SpMM, GEMM, the fused launch, dW, dA, the transposed SpMM, Mᵀ) and compares
every result with the oracle at the stated 1e-5 tolerance."""
import os
import shutil
import subprocess

import pytest

from _util import ROOT, load_c_oracle

pytestmark = pytest.mark.gpu


def test_c_host_drives_the_layer_through_the_abi(tmp_path):
    gcc = shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    load_c_oracle()                                            # builds oracle/libtmgcn_ref.so if missing
    lib_dir = os.path.join(ROOT, "tm-gcn_amd")
    assert os.path.exists(os.path.join(lib_dir, "libtmgcn_hip.so")), "build the HIP library first"
    exe = str(tmp_path / "abi_driver")
    # a C compiler, the HIP runtime's C API, our header and library — nothing else
    subprocess.check_call([gcc, "-std=c99", "-O1", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"),
                           os.path.join(ROOT, "tests", "abi_driver", "driver.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + lib_dir, "-ltmgcn_hip",
                           "-L" + os.path.join(ROOT, "oracle"), "-ltmgcn_ref",
                           "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                           "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_driver OK" in r.stdout, r.stdout
