#!/usr/bin/env python3
"""Digest of a rocprofv3 --memory-copy-trace --kernel-trace run of tools/pipe_trace.py: per copy direction the count,
bytes and achieved GB/s, and how much of each large device-to-host copy ran while kernels of the NEXT step were
executing (overlap = time covered by kernel executions / copy duration).
usage: tools/copy_trace_summary.py <memory_copy_trace.csv> <kernel_trace.csv>"""
import csv
import sys

copies = list(csv.DictReader(open(sys.argv[1])))
kernels = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[2]))]
kernels.sort()
by = {}
for r in copies:
    d = r.get("Direction", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = int(r.get("Bytes", r.get("Size", 0)) or 0)
    by.setdefault(d, []).append((s, e, n))
print("# rocprofv3 --memory-copy-trace digest (tools/pipe_trace.py)")
print("%-28s %6s %12s %10s %10s" % ("direction", "count", "MB", "ms", "GB/s"))
for d, v in by.items():
    ms = sum(e - s for s, e, n in v) / 1e6
    mb = sum(n for s, e, n in v) / 1e6
    print("%-28s %6d %12.1f %10.2f %10.1f" % (d, len(v), mb, ms, mb / max(ms, 1e-9)))
big = sorted([c for d, v in by.items() if "DEVICE_TO_HOST" in d.upper() or "D2H" in d.upper() for c in v if c[2] > 50e6])
print("\n# device-to-host copies > 50 MB: share of their duration during which a kernel was executing")
for s, e, n in big:
    cov = 0
    for ks, ke in kernels:
        if ke <= s:
            continue
        if ks >= e:
            break
        cov += min(e, ke) - max(s, ks)
    print("copy %7.1f MB  %7.2f ms  %6.1f GB/s  kernels busy during %5.1f %% of it" % (n / 1e6, (e - s) / 1e6, n / (e - s) * 1e3 / 1e3, 100.0 * cov / (e - s)))
