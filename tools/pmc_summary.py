#!/usr/bin/env python3
"""tools/pmc_summary.py — per-kernel summary of rocprofv3 counter passes (one directory per pass, CSV output).

    python tools/pmc_summary.py <dir_sq> <dir_fetch> <dir_write> [--elements N]

Prints, per kernel name: dispatches, mean duration (kernel trace of the SQ pass), HBM read bytes (FETCH_SIZE, in KiB,
DOUBLED on gfx950 for 16-byte-per-lane streaming reads as MI355X_MICROARCH.md prescribes), written bytes (WRITE_SIZE),
and the SQ shares: active-VALU / wave cycles, waiting (s_waitcnt, barrier) / wave cycles, VALU instructions per wave.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def load_counters(d):
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def load_durations(d):
    out = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out


def mean(v):
    return sum(v) / len(v) if v else float("nan")


def main():
    sq, fe, wr = (load_counters(a) for a in sys.argv[1:4])
    dur = load_durations(sys.argv[1])
    print(f"{'kernel':70s} {'n':>4s} {'us':>8s} {'read MB':>9s} {'write MB':>9s} {'VALU act%':>9s} {'wait%':>7s} {'VALU/wave':>10s} {'VMEM rd/wave':>12s}")
    for k in sorted(sq, key=lambda k: -mean(dur.get(k, [0]))):
        if not (k.startswith(("void dmxq", "dmxq")) or any(t in k for t in ("binary_range_bf16_kernel", "fused_cast_generic_kernel", "float_range_bf16_kernel"))):
            continue
        c = sq[k]
        wc = mean(c.get("SQ_WAVE_CYCLES", []))
        waves = mean(c.get("SQ_WAVES", []))
        rd = 2 * mean(fe.get(k, {}).get("FETCH_SIZE", [])) * 1024 / 1e6
        ww = mean(wr.get(k, {}).get("WRITE_SIZE", [])) * 1024 / 1e6
        short = k.replace("void ", "")[:70]
        print(f"{short:70s} {len(dur.get(k, [])):4d} {mean(dur.get(k, [])):8.2f} {rd:9.2f} {ww:9.2f} "
              f"{100 * mean(c.get('SQ_ACTIVE_INST_VALU', [])) / wc:9.1f} {100 * mean(c.get('SQ_WAIT_ANY', [])) / wc:7.1f} "
              f"{mean(c.get('SQ_INSTS_VALU', [])) / waves:10.1f} {mean(c.get('SQ_INSTS_VMEM_RD', [])) / waves:12.1f}")


if __name__ == "__main__":
    main()
