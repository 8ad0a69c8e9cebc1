#!/usr/bin/env python
"""profiles/MANIFEST.json — provenance of every committed capture under profiles/ (VERDICT r2 item 2c).

The GPU box has no .git, so bench.py cannot ask git which build a replayed number (PMC traffic, rocprofv3 kernel
duration, box peaks) belongs to; it reads this manifest instead.  Per file: sha256 of the content, the commit that last
touched it, that commit's date, and (for the files bench.py replays) the role.  Run after committing new captures:

    python tools/write_manifest.py && git add profiles/MANIFEST.json && git commit
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def git(*a):
    return subprocess.check_output(("git", "-C", ROOT) + a, stderr=subprocess.DEVNULL).decode().strip()


def main():
    sys.path.insert(0, ROOT)
    import bench
    roles = {bench.PMC_FILE: "roofline.traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)",
             bench.STATS_FILE: "roofline.from_profiles.rocprof_us_per_launch (rocprofv3 --kernel-trace --stats)",
             bench.MFMA_FILE: "fc_mfma_utilisation (rocprofv3 --pmc MfmaUtil)",
             "profiles/r01_box.json": "peak_measured (tools/exp/box_probe.hip on the MI355X box)"}
    files = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*"))):
        rel = os.path.relpath(f, ROOT)
        if not os.path.isfile(f) or rel == "profiles/MANIFEST.json":
            continue
        try:
            commit, date = git("log", "-1", "--format=%h %cI", "--", rel).split()
        except Exception:
            commit, date = None, None
        e = {"sha256": hashlib.sha256(open(f, "rb").read()).hexdigest(), "commit": commit, "date": date}
        if rel in roles:
            e["replayed_by_bench_as"] = roles[rel]
        files[rel] = e
    out = {"note": "provenance of the committed captures; bench.py reads `git` from here when the tree has no .git (GPU box)",
           "git": git("log", "-1", "--format=%h", "--", "profiles"), "files": files}
    with open(os.path.join(ROOT, "profiles", "MANIFEST.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("profiles/MANIFEST.json: %d files, profiles/ last touched by %s" % (len(files), out["git"]))


if __name__ == "__main__":
    main()
