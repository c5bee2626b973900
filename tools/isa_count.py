#!/usr/bin/env python3
"""tools/isa_count.py <file.s> [substring ...] -- per kernel of a hipcc -S listing: instruction totals by class (VALU, transcendental,
SALU, branches, vector memory), for kernels whose DEMANGLED name contains every given substring.  A static count of the whole
kernel body (all paths), good for comparing two forms of one kernel."""
import re
import subprocess
import sys


def main():
    s = open(sys.argv[1]).read()
    subs = sys.argv[2:]
    labels = re.findall(r"^(_Z\w+):", s, flags=re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(labels), capture_output=True, text=True).stdout.splitlines()
    for lab, d in zip(labels, dem):
        if not all(t in d for t in subs):
            continue
        i = s.index("\n" + lab + ":")
        j = s.index("s_endpgm", i)
        ins = [ln.strip().split()[0] for ln in s[i:j].splitlines() if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((".", ";"))]
        cnt = lambda f: sum(1 for x in ins if f(x))
        print(re.sub(r"\(.*", "", d.replace("void ", "").replace("dmxq::", ""))[:160])
        print(f"   total {len(ins)}  valu {cnt(lambda x: x.startswith('v_'))}  transcendental {cnt(lambda x: re.match(r'v_(exp|rcp|log|rsq|sqrt|sin|cos)', x))}"
              f"  salu {cnt(lambda x: x.startswith('s_'))}  branches {cnt(lambda x: x.startswith('s_cbranch'))}"
              f"  vmem {cnt(lambda x: x.startswith(('global_', 'buffer_', 'flat_')))}  lds {cnt(lambda x: x.startswith('ds_'))}")


if __name__ == "__main__":
    main()
