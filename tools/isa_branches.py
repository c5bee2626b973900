#!/usr/bin/env python3
"""tools/isa_branches.py — per kernel of an ISA listing (hipcc -S --cuda-device-only): instruction count, scalar branches,
v_div / transcendental counts.  Scalar branches inside the per-vector code keep the scheduler from interleaving independent
chains; this lists the kernels where they are dense.

    python tools/isa_branches.py /tmp/asm/ew2.s [--grep "2, 2"]
"""
import re
import subprocess
import sys


def main():
    args = sys.argv[1:]
    pat = None
    if "--grep" in args:
        i = args.index("--grep")
        pat = args[i + 1]
        del args[i:i + 2]
    rows = []
    for path in args:
        cur = None
        for line in open(path, errors="replace"):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = {"name": m.group(1), "n": 0, "br": 0, "div": 0, "trans": 0, "vmem": 0}
                rows.append(cur)
                continue
            if cur is None:
                continue
            t = line.strip()
            if t.startswith("s_endpgm"):
                cur = None
                continue
            if not t or t[0] in ".;" or t.endswith(":"):
                continue
            cur["n"] += 1
            op = t.split()[0]
            if op.startswith("s_cbranch"):
                cur["br"] += 1
            elif op.startswith("v_div_"):
                cur["div"] += 1
            elif op.startswith(("v_rcp", "v_exp", "v_log", "v_sqrt", "v_rsq")):
                cur["trans"] += 1
            elif op.startswith(("global_load", "global_store", "buffer_load", "buffer_store")):
                cur["vmem"] += 1
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'instr':>6} {'branch':>6} {'per100':>6} {'v_div':>5} {'trans':>5} {'vmem':>5}  kernel")
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n.replace("void ", "").replace("dmxq::", ""))
        if pat and pat not in n:
            continue
        print(f"{r['n']:>6} {r['br']:>6} {100.0 * r['br'] / max(1, r['n']):>6.1f} {r['div']:>5} {r['trans']:>5} {r['vmem']:>5}  {n[:150]}")


if __name__ == "__main__":
    main()
