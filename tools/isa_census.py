"""Static census of the product's gfx950 code (no GPU needed): per kernel — VGPRs, LDS bytes, scratch, the compiler's occupancy estimate, and the
histogram of `s_waitcnt vmcnt(N)` beside the counts of global loads / MFMAs / barriers.  Two findings of round 4 came out of this listing:
a prefetch ring whose guarded load made hipcc wait with vmcnt(3 .. 0) — draining the loads just issued — in every chunk of the weight-gradient
kernels, and a dgrad epilogue with sixteen dependent load -> vmcnt(0) -> store round trips.
    python tools/isa_census.py [sdqn_kernels_bt.hip ...] [-k substring]        (default: every translation unit of simple_dqn_amd/csrc)"""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "simple_dqn_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-S"]


def demangle(names):
    try:
        out = subprocess.run([os.environ.get("CXXFILT", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, out)) if len(out) == len(names) else {n: n for n in names}
    except OSError:
        return {n: n for n in names}


def census_rows(src, extra_flags=()):
    """Per-kernel records of one translation unit: name (mangled), vgpr, agpr, lds, scratch, occ, loads, mfma, barriers, vmcnt (Counter)."""
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call([HIPCC] + FLAGS + list(extra_flags) + ["-o", asm, os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
        txt = open(asm).read()
    rows = []
    # "; -- Begin function <name>" ... code ... "; -- End function" ... "; Kernel info:" comments, up to the next Begin marker
    for m in re.finditer(r"; -- Begin function (\S+)\n(.*?); -- End function\n(.*?)(?=; -- Begin function|\Z)", txt, re.S):
        name, body, tail = m.group(1), m.group(2), m.group(3)
        v = re.search(r"; NumVgprs: (\d+)", tail)
        if not v or "; Kernel info:" not in tail:
            continue
        g = lambda pat: int(re.search(pat, tail).group(1))
        rows.append(dict(name=name, vgpr=int(v.group(1)), agpr=g(r"; NumAgprs: (\d+)"), lds=g(r"; LDSByteSize: (\d+)"), scratch=g(r"; ScratchSize: (\d+)"), occ=g(r"; Occupancy: (\d+)"),
                         loads=len(re.findall(r"\bglobal_load|\bbuffer_load", body)), mfma=len(re.findall(r"\bv_mfma", body)),
                         barriers=len(re.findall(r"\bs_barrier", body)), vmcnt=Counter(int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body))))
    return rows


def census(src, key=None):
    rows = census_rows(src)
    names = demangle([r["name"] for r in rows])
    for r in rows:
        nm = names[r["name"]].replace("sdqn::", "")
        if key and key not in nm:
            continue
        w = " ".join("%d:%d" % (k, r["vmcnt"][k]) for k in sorted(r["vmcnt"]))
        print("%-150s vgpr %3d+%-3d occ %d lds %6d scratch %d | loads %3d mfma %3d barriers %2d | vmcnt(N):count  %s" % (
            nm[:150], r["vgpr"], r["agpr"], r["occ"], r["lds"], r["scratch"], r["loads"], r["mfma"], r["barriers"], w))


if __name__ == "__main__":
    args = sys.argv[1:]
    key = None
    if "-k" in args:
        key = args[args.index("-k") + 1]; del args[args.index("-k"):args.index("-k") + 2]
    for src in (args or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))):
        print("== " + src, flush=True)
        census(src, key)
