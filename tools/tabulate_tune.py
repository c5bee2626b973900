#!/usr/bin/env python3
"""tools/tabulate_tune.py <prefix> <size> [<size> ...] -- side-by-side table (median us, % of 8 TB/s) of tools/tune_stream / tune_lastdim
outputs <prefix><size>.txt in the current directory; the best geometry of every op per size is marked with *."""
import re
import sys

pre, sizes = sys.argv[1], sys.argv[2:]
tab, order = {}, []
for z in sizes:
    for l in open(f"{pre}{z}.txt"):
        m = re.match(r"(\S+)\s+(\S*\d+x\d+|T\d+ R\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)%", l)
        if m:
            k = (m.group(1), m.group(2))
            if k not in tab:
                tab[k] = {}
                order.append(k)
            tab[k][z] = float(m.group(4))
best = {}
for (op, g), d in tab.items():
    for z, v in d.items():
        if v < best.get((op, z), 1e9):
            best[(op, z)] = v
print("%-16s" % "variant" + "".join("%10s" % z for z in sizes))
for k in order:
    print("%-7s %-8s" % k + "".join(("%9.2f%s" % (tab[k][z], "*" if tab[k][z] == best[(k[0], z)] else " ")) if z in tab[k] else "%10s" % "-" for z in sizes))
