#!/usr/bin/env python3
"""Per-kernel SQ counter digest from two rocprofv3 --pmc passes (A: wave/busy/active/wait cycles, B: LDS conflicts and
FP64 instruction mix), averaged per launch, with the derived figures DESIGN.md quotes.

usage: tools/sq_counters_summary.py <passA_counter_collection.csv> <passB_counter_collection.csv> <kernel_trace.csv>
Quad-cycle counters (SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_*) are reported by the hardware in units of 4 clocks.
"""
import collections
import csv
import re
import sys


def per_launch(path):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(path)):
        mt = re.search(r"(\w+_kernel)\b", r["Kernel_Name"])
        if not mt or "at::native" in r["Kernel_Name"]:
            continue
        k = mt.group(1)
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
    return {k: {c: v / n[k][c] for c, v in cs.items()} for k, cs in tot.items()}


def durations(path):
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        mt = re.search(r"(\w+_kernel)\b", r["Kernel_Name"])
        if not mt or "at::native" in r["Kernel_Name"]:
            continue
        tot[mt.group(1)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        n[mt.group(1)] += 1
    return {k: tot[k] / n[k] for k in tot}


def main():
    a, b, dur = per_launch(sys.argv[1]), per_launch(sys.argv[2]), durations(sys.argv[3])
    print("# SQ counters per launch (rocprofv3 --pmc, two passes)%s; durations under the profiler in microseconds" % ((", " + sys.argv[4]) if len(sys.argv) > 4 else ""))
    print("# wave-level shares: fraction of a resident wave's lifetime; fp64_TFLOPs = (2*FMA + ADD + MUL + TRANS) * 64 lanes / time")
    print("%-22s %9s %9s %9s %9s %9s %10s %10s %11s" % ("kernel", "dur_us", "valu%", "lds%", "wait%", "nowait%",
                                                       "lds_confl%", "fp64_inst%", "fp64_TFLOPs"))
    for k in sorted(dur, key=lambda x: -dur[x]):
        if k not in a or k not in b or dur[k] < 100:
            continue
        ca, cb = dict(b[k], **a[k]), dict(a[k], **b[k])  # a counter may have been collected in either pass
        wc = ca.get("SQ_WAVE_CYCLES", 0) or 1
        fl = 2 * cb.get("SQ_INSTS_VALU_FMA_F64", 0) + cb.get("SQ_INSTS_VALU_ADD_F64", 0) + cb.get("SQ_INSTS_VALU_MUL_F64", 0) + \
            cb.get("SQ_INSTS_VALU_TRANS_F64", 0)
        f64i = cb.get("SQ_INSTS_VALU_FMA_F64", 0) + cb.get("SQ_INSTS_VALU_ADD_F64", 0) + cb.get("SQ_INSTS_VALU_MUL_F64", 0) + \
            cb.get("SQ_INSTS_VALU_TRANS_F64", 0)
        print("%-22s %9.0f %8.1f%% %8.1f%% %8.1f%% %8.1f%% %9.1f%% %9.1f%% %11.2f" % (
            k, dur[k], 100 * ca.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * ca.get("SQ_ACTIVE_INST_LDS", 0) / wc,
            100 * ca.get("SQ_WAIT_ANY", 0) / wc, 100 * ca.get("SQ_WAIT_INST_ANY", 0) / wc,
            100 * cb.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, cb.get("SQ_LDS_IDX_ACTIVE", 0)),
            100 * f64i / max(1.0, ca.get("SQ_INSTS_VALU", 0)), fl * 64 / (dur[k] * 1e-6) / 1e12))


if __name__ == "__main__":
    main()
