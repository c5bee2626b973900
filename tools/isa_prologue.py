#!/usr/bin/env python3
"""tools/isa_prologue.py <file.s> ... -- per kernel of a hipcc -S listing: what runs BEFORE the first streaming (non-temporal) load.
Round 3 found two kernels that waited on parameter loads and did a few hundred VALU operations of parameter arithmetic before
requesting their first data (lastdim_kernel, layernorm_wave_kernel): a full memory round trip in front of every workgroup.  This lists,
for every kernel, the VALU count, the vector loads and the `s_waitcnt vmcnt` waits that precede its first `nt` load, and flags kernels
that wait on vector memory before it."""
import re
import subprocess
import sys


def main():
    rows = []
    for path in sys.argv[1:]:
        s = open(path).read()
        labels = re.findall(r"^(_Z\w+):", s, flags=re.M)
        if not labels:
            continue
        dem = subprocess.run(["c++filt"], input="\n".join(labels), capture_output=True, text=True).stdout.splitlines()
        for lab, d in zip(labels, dem):
            i = s.index("\n" + lab + ":")
            j = s.find("s_endpgm", i)
            if j < 0:
                continue
            ins = [ln.strip() for ln in s[i:j].splitlines() if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((".", ";"))]
            first_nt = next((k for k, x in enumerate(ins) if x.startswith(("global_load", "buffer_load")) and x.rstrip().endswith(" nt")), None)
            if first_nt is None:
                continue
            pre = ins[:first_nt]
            valu = sum(1 for x in pre if x.startswith("v_"))
            loads = sum(1 for x in pre if x.startswith(("global_load", "buffer_load")))
            waits = [x for x in pre if x.startswith("s_waitcnt") and "vmcnt" in x]
            branches = sum(1 for x in pre if x.startswith("s_cbranch"))
            name = re.sub(r"\(.*", "", d.replace("void ", "").replace("dmxq::", ""))
            rows.append((len(waits) > 0 and loads > 0, valu, loads, len(waits), branches, name[:150]))
    rows.sort(key=lambda r: (-int(r[0]), -r[1]))
    print(f"{'waits?':6s} {'VALU':>5s} {'loads':>5s} {'vmcnt':>5s} {'br':>3s}  kernel (before its first non-temporal load)")
    for w, valu, loads, nw, br, name in rows:
        print(f"{'WAIT' if w else '':6s} {valu:5d} {loads:5d} {nw:5d} {br:3d}  {name}")


if __name__ == "__main__":
    main()
