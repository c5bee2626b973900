#!/usr/bin/env python3
"""tools/kernel_resources.py — VGPR / SGPR / scratch / occupancy of every kernel in one source file, from the compiler's own
-Rpass-analysis=kernel-resource-usage remarks (no GPU needed).

    python tools/kernel_resources.py dmx-compressor_amd/csrc/elementwise.hip [-DDMXQ_EW_PART=2] [--grep FixedOp]
"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-fast-math", "-ffp-contract=off", "-fno-gpu-flush-denormals-to-zero"]


def main():
    args = sys.argv[1:]
    pat = None
    if "--grep" in args:
        i = args.index("--grep")
        pat = args[i + 1]
        del args[i:i + 2]
    src, extra = args[0], args[1:]
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" ")[0]] = v
    names = subprocess.run(["c++filt"], input="\n".join(x["name"] for x in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'VGPR':>5} {'SGPR':>5} {'scratch':>7} {'LDS':>6} {'occ':>3}  kernel")
    for x, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n.replace("void ", "").replace("dmxq::", ""))
        if pat and pat not in n:
            continue
        print(f"{x.get('VGPRs', '?'):>5} {x.get('TotalSGPRs', '?'):>5} {x.get('ScratchSize', '?'):>7} {x.get('LDS', '?'):>6} {x.get('Occupancy', '?'):>3}  {n}")


if __name__ == "__main__":
    main()
