"""Build a variant of libpairnet_hip.so under tools/gpubin/ with extra flags for all or some
sources:  python tools/build_variant.py NAME "FLAGS" [source,source,...]
(sources not listed are compiled with the library's own flags).  Run the package against it with
PAIRNET_LIB=tools/gpubin/libpairnet_NAME.so."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import build as B

name, extra = sys.argv[1], sys.argv[2].split()
only = sys.argv[3].split(",") if len(sys.argv) > 3 else B.SOURCES
out = os.path.join(B.ROOT, "tools", "gpubin")
objdir = os.path.join(out, "obj_" + name)
os.makedirs(objdir, exist_ok=True)


def cc(src):
    obj = os.path.join(objdir, src + ".o")
    flags = B.FLAGS + (extra if src in only else [])
    r = subprocess.run([B._hipcc()] + flags + ["-c", os.path.join(B.CSRC, src + ".hip"), "-o", obj],
                       capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr)
    return obj


with ThreadPoolExecutor(max_workers=4) as ex:
    objs = list(ex.map(cc, B.SOURCES))
lib = os.path.join(out, "libpairnet_%s.so" % name)
r = subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib],
                   capture_output=True, text=True)
if r.returncode:
    raise SystemExit(r.stderr)
print(lib)
