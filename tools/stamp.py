#!/usr/bin/env python3
"""tools/stamp.py — WHICH library a file under profiles/ was measured on.

    python tools/stamp.py --write     build container, before `gpurun`: records `git rev-parse HEAD` (+ "dirty" when the tree has
                                      uncommitted changes) and the SHA-256 of dmx-compressor_amd/lib/{libdmxq.so,dmxq_torch.so} in
                                      tools/.build_stamp.json (git-ignored, travels with the gpurun snapshot: the GPU box has no .git)
    python tools/stamp.py --header    GPU box: one comment line for the top of a profile file -- the recorded commit, the SHA-256 of the
                                      library files actually present (recomputed), and whether they are the recorded ones
    python tools/stamp.py --json      the same as a JSON object (bench.py puts it into its line as `build`)
"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMP = os.path.join(ROOT, "tools", ".build_stamp.json")
LIBS = [os.path.join(ROOT, "dmx-compressor_amd", "lib", n) for n in ("libdmxq.so", "dmxq_torch.so")]


def sha(path):
    h = hashlib.sha256()
    try:
        with open(path, "rb") as f:
            for b in iter(lambda: f.read(1 << 20), b""):
                h.update(b)
    except OSError:
        return None
    return h.hexdigest()


def write():
    commit = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=ROOT, text=True).strip()
    dirty = subprocess.run(["git", "diff", "--quiet", "HEAD", "--", "dmx-compressor_amd", "include"], cwd=ROOT).returncode != 0
    d = {"commit": commit, "dirty": dirty, "written": time.strftime("%Y-%m-%dT%H:%MZ", time.gmtime()),
         "sha256": {os.path.basename(p): sha(p) for p in LIBS}}
    json.dump(d, open(STAMP, "w"), indent=1)
    return d


def current():
    try:
        rec = json.load(open(STAMP))
    except OSError:
        rec = None
    now = {os.path.basename(p): sha(p) for p in LIBS}
    if rec is None and os.path.isdir(os.path.join(ROOT, ".git")):
        rec = write()
    return {"commit": (rec or {}).get("commit"), "dirty": (rec or {}).get("dirty"), "libdmxq_sha256": now["libdmxq.so"],
            "dmxq_torch_sha256": now["dmxq_torch.so"], "matches_stamp": bool(rec) and rec["sha256"] == now}


if __name__ == "__main__":
    if "--write" in sys.argv:
        print(json.dumps(write()))
    elif "--json" in sys.argv:
        print(json.dumps(current()))
    else:
        c = current()
        print(f"# build: commit {c['commit']}{' + uncommitted changes' if c['dirty'] else ''}, libdmxq.so sha256 {c['libdmxq_sha256']}, "
              f"dmxq_torch.so sha256 {(c['dmxq_torch_sha256'] or '')[:16]} ({'the stamped build' if c['matches_stamp'] else 'NOT the stamped build'}); "
              f"measured {time.strftime('%Y-%m-%dT%H:%MZ', time.gmtime())}")
