"""Recipe for oracle/_ref/ — test/bench infrastructure, NOT product code.

The only part of the reference that runs without Neon is src/replay_memory.py.  It is Python, so
"building" it means byte-compiling it FROM WHERE IT LIES under /root/reference into oracle/_ref/replay_memory.pyc:
no reference source is copied into the repository, oracle/_ref/ is git-ignored (it stays out of history) but travels
to the GPU box with the snapshot like the in-tree .so files, so that bench.py can time the reference's own
ReplayMemory.getMinibatch() on the GPU box's host cores where /root/reference does not exist.
The .pyc is tied to this image's CPython (3.10); oracle/ref_loader.py refuses a mismatching one.

    python oracle/build_ref.py          (also run by __graft_entry__.build() when /root/reference is present)
"""
import os
import py_compile
import sys

REF_SRC = "/root/reference/src/replay_memory.py"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
OUT = os.path.join(OUT_DIR, "replay_memory.pyc")


def build():
    if not os.path.exists(REF_SRC):
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    # UNCHECKED_HASH: the .pyc does not depend on the (absent, on the GPU box) source file's mtime
    py_compile.compile(REF_SRC, cfile=OUT, doraise=True, invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    return OUT


if __name__ == "__main__":
    out = build()
    print(out if out else "reference not present (%s): nothing built" % REF_SRC)
    sys.exit(0)
