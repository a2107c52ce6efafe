"""ctypes binding of the plain-C oracle (oracle/tmgcn_ref.c -> oracle/libtmgcn_ref.so).
TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's checker
legs (cpu_baseline, verify) may import this; nothing under tm-gcn_amd/ does."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load(build_if_stale: bool = True):
    """The loaded library with argument types set.  Rebuilt with the committed Makefile when the
    .so is missing or older than its source (the GPU box receives the prebuilt file)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path, src = os.path.join(HERE, "libtmgcn_ref.so"), os.path.join(HERE, "tmgcn_ref.c")
    override = os.environ.get("TMGCN_REF_LIB")      # tests/sanitize: the -fsanitize=address,undefined build of the same source
    if override:
        path, build_if_stale = override, False
    def stale():
        return not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path))

    if stale() and build_if_stale:
        # several processes may get here at once (the ranks of a multi-GPU bench after a fresh checkout): one builds,
        # the others wait on the lock and find the library fresh; the build runs without the profiler / preload
        # variables a GPU rank may carry
        import fcntl
        with open(os.path.join(HERE, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if stale():
                env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
                subprocess.check_call(["make", "-C", HERE], env=env)
    lib = C.CDLL(path)
    p, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.ref_mtransform.argtypes = [p, C.c_int, C.c_int, p, p, i64]
    lib.ref_mtransform_rows.argtypes = [p, C.c_int, C.c_int, C.c_int, C.c_int, p, p, i64]
    lib.ref_spmm.argtypes = [p, p, p, p, p, i64, i32, i32]
    lib.ref_gemm.argtypes = [p, p, p, i64, i32, i32, i32, i64, i64]
    lib.ref_gemm_dw.argtypes = [p, p, p, i64, i32, i32, i64]
    for f in (lib.ref_mtransform, lib.ref_mtransform_rows, lib.ref_spmm, lib.ref_gemm, lib.ref_gemm_dw):
        f.restype = None
    _LIB = lib
    return lib


def cptr(t):
    return C.c_void_p(t.data_ptr())
