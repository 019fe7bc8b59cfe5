"""Build libpairnet_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so sits
next to the sources so that it travels with the repo snapshot to the GPU box."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpairnet_hip.so")
SOURCES = ["gemm", "gemm_ln", "gemm_s3", "stem", "winograd", "ffn", "norm", "msda", "resize", "attn", "ppn", "postproc", "swin", "preprocess", "detr", "loss", "grad", "optim"]
# No packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) anywhere in the
# library: while waves of the bf16-MFMA GEMM (csrc/gemm_s3.hip: dense MFMAs + LDS-DMA) are
# resident on a CU, such instructions in OTHER kernels' waves were measured to return wrong
# results on this stack (k_msda: 31-34 of 40 launches wrong beside it, 0 of 40 without them;
# reproducer from a clean checkout: tools/coresidency_probe.py, profiles/r06_coresidency.txt;
# history: labnotes R5.12, R6.3).  The target feature is switched off rather than the SLP
# vectoriser: not one v_pk_*_f32 is left (tests/test_boundary.py checks the built library).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build libpairnet_hip.so")
    return exe


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(ROOT, "include", "pairnet_hip.h"))
    deps.append(os.path.abspath(__file__))          # the flags are part of the build
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link the shared library."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = _hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    objdir = os.path.join(LIB_DIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def cc(name):
        obj = os.path.join(objdir, name + ".o")
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, name + ".hip"), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (name, r.stderr))
        if verbose and r.stderr:
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs
                       + ["-o", LIB_PATH], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
