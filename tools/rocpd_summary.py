#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite result (ROCm 7.2 default output) into the kernel-stats text that gets
committed under profiles/.  usage: tools/rocpd_summary.py <results.db> [> profiles/xxx.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                   "max(lds_size), max(grid_x*grid_y*grid_z) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)")
print("%-64s %7s %12s %10s %10s %10s %8s %6s %10s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "lds", "grid"))
for name, calls, total, avg, mn, mx, lds, grid in rows:
    short = name if len(name) <= 64 else name[:61] + "..."
    print("%-64s %7d %12.1f %10.1f %10.1f %10.1f %7.2f%% %6d %10d" % (short, calls, total / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                    100.0 * total / tot, lds or 0, grid or 0))
try:
    reg = cur.execute("select k.name, s.arch_vgpr_count, s.accum_vgpr_count, s.sgpr_count, s.group_segment_size from "
                      "(select distinct name, kernel_id from kernels) k join kernel_symbols s on s.kernel_id = k.kernel_id").fetchall()
    print("\n# registers / static LDS per kernel symbol")
    for name, v, a, s, g in reg:
        short = name if len(name) <= 64 else name[:61] + "..."
        print("%-64s vgpr=%s agpr=%s sgpr=%s static_lds=%s" % (short, v, a, s, g))
except Exception as e:  # schema differences between ROCm releases
    print("# (register table unavailable: %s)" % e)
