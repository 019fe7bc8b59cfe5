"""Count packed-fp32 instructions (v_pk_*_f32) per embedded gfx950 code object of a built
libpairnet_hip.so (LABNOTES R5.12: their results were measured wrong while a bf16-MFMA kernel of
another stream was resident on the CU).   python tools/check_packed_fp32.py [path/to/lib.so]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_objects(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
        data = open(fat, "rb").read()
        for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data):
            st = m.start()
            n = struct.unpack_from("<Q", data, st + 24)[0]
            off = st + 32
            for _ in range(n):
                o, sz, tl = struct.unpack_from("<QQQ", data, off)
                off += 24
                triple = data[off:off + tl].decode()
                off += tl
                if "gfx950" in triple and sz:
                    co = os.path.join(tmp, "dev_%d.co" % st)
                    open(co, "wb").write(data[st + o:st + o + sz])
                    yield subprocess.run([LLVM + "/llvm-objdump", "-d", "--mcpu=gfx950", co],
                                         capture_output=True, text=True, check=True).stdout


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pair-net_amd", "lib", "libpairnet_hip.so")
    total = kernels = 0
    for asm in code_objects(lib):
        per = {}
        cur = None
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1)
            elif cur and re.search(r"v_pk_[a-z0-9]+_f32", line):
                per[cur] = per.get(cur, 0) + 1
        kernels += asm.count("s_endpgm")
        for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:6]:
            print("%6d  %s" % (v, k[:100]))
        total += sum(per.values())
    print("%d packed-fp32 instructions in %d kernels of %s" % (total, kernels, lib))


if __name__ == "__main__":
    main()
