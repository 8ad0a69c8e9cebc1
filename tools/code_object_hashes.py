"""sha256 of every gfx950 code object embedded in a built library (the device code of each translation unit), keyed by the kernels it
defines: `python tools/code_object_hashes.py simple_dqn_amd/libsdqn_hip.so [> file]`.  Two builds whose outputs are equal run the same
kernels bit for bit — the check behind a refactoring that is supposed to move host code only.
`--kernels` lists one line per kernel instead (sha256 of the function's own machine code): when a refactoring also REMOVES kernels that
nothing can launch any more, every surviving kernel must keep its line (`--compare old.txt new.txt` says which lines differ)."""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, want_kernels=False):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fatbin")
        subprocess.check_call([OBJCOPY, "--dump-section", ".hip_fatbin=" + fat, lib])
        data = open(fat, "rb").read()
        pos, out = 0, []
        while True:
            pos = data.find(MAGIC, pos)
            if pos < 0:
                break
            n = struct.unpack_from("<Q", data, pos + 24)[0]
            p = pos + 32
            for _ in range(n):
                off, size, tlen = struct.unpack_from("<QQQ", data, p)
                triple = data[p + 24:p + 24 + tlen].decode()
                p += 24 + tlen
                if "gfx950" in triple and size:
                    blob = data[pos + off:pos + off + size]
                    co = os.path.join(td, "co")
                    open(co, "wb").write(blob)
                    syms = subprocess.run([READELF, "--symbols", "--wide", co], capture_output=True, text=True).stdout
                    kernels = sorted(l.split()[-1] for l in syms.splitlines() if " FUNC " in l and " GLOBAL " in l)
                    per = {}
                    if want_kernels:
                        secs = subprocess.run([READELF, "--sections", "--wide", co], capture_output=True, text=True).stdout
                        text = [l.split() for l in secs.splitlines() if " .text " in l][0]
                        k = text.index(".text")
                        t_addr, t_off = int(text[k + 2], 16), int(text[k + 3], 16)
                        for l in syms.splitlines():
                            f = l.split()
                            if " FUNC " in l and " GLOBAL " in l:
                                addr, size = int(f[1], 16), int(f[2])
                                per[f[-1]] = hashlib.sha256(blob[t_off + addr - t_addr:t_off + addr - t_addr + size]).hexdigest()
                    out.append((hashlib.sha256(blob).hexdigest(), len(blob), kernels, per))
            pos += len(MAGIC)
        return out


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        old = dict(l.split()[::-1] for l in open(sys.argv[2]) if l.strip())
        new = dict(l.split()[::-1] for l in open(sys.argv[3]) if l.strip())
        changed = sorted(k for k in new if k in old and old[k] != new[k])
        print("kernels: %d before, %d after; removed %d, added %d, CHANGED %d" % (len(old), len(new), len(set(old) - set(new)), len(set(new) - set(old)), len(changed)))
        for k in changed:
            print("changed:", k)
        for k in sorted(set(new) - set(old)):
            print("added:", k)
        sys.exit(1 if changed else 0)
    if sys.argv[1] == "--kernels":
        for r in code_objects(sys.argv[2], True):
            for k, h in sorted(r[3].items()):
                print(h, k)
        sys.exit(0)
    rows = code_objects(sys.argv[1])
    for h, n, kernels, _ in sorted(rows, key=lambda r: r[2][:1]):
        print("%s %8d bytes  %3d kernels  first: %s" % (h, n, len(kernels), kernels[0][:80] if kernels else "-"))
    print("all: %s" % hashlib.sha256("".join(sorted(r[0] for r in rows)).encode()).hexdigest())
