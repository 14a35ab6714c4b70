#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately,
as /opt/skills/guides/MI355X_MICROARCH.md prescribes), written in the table bench.py's `roofline.traffic` reads.

usage: tools/pmc_traffic_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <frames_per_launch>
           [label] [fs=16000] [fft=1024] [requiem=0] [out_hop_scale=1]  > profiles/hbm_traffic_cfg2_latest.txt
   e.g. config 4: ... 128064 "config 4 (64 x 10 s, Requiem)" fs=16000 fft=1024 requiem=1
        config 5: ... 192016 "config 5 (16 x 60 s, 48 kHz)" fs=48000 fft=2048 out_hop_scale=2

rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (the guide's
HBM section), so corrected_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Values are averaged over the launches of each
kernel in the run.  algorithmic_MB uses bench.algo_bytes_per_frame (SURVEY 8(d) components) for the given rate, FFT size,
aperiodicity format (requiem=1: band values instead of the 513-bin row) and output hop scale (scale_duration).
"""
import collections
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))


def per_launch(path, counter):
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        mt = re.search(r"(\w+_kernel)\b", r["Kernel_Name"])
        if not mt or "at::native" in r["Kernel_Name"]:
            continue
        name = mt.group(1)
        tot[name] += float(r["Counter_Value"])
        n[name] += 1
    return {k: tot[k] / n[k] for k in tot}, n


def main():
    fetch, nf = per_launch(sys.argv[1], "FETCH_SIZE")
    write, _ = per_launch(sys.argv[2], "WRITE_SIZE")
    frames = int(sys.argv[3])
    label = sys.argv[4] if len(sys.argv) > 4 and "=" not in sys.argv[4] else "config 2 (64 x 10 s)"
    opt = dict(a.split("=", 1) for a in sys.argv[4:] if "=" in a)
    fs, fft = int(opt.get("fs", 16000)), int(opt.get("fft", 1024))
    import bench
    algo, _ = bench.algo_bytes_per_frame(fs, fft, float(opt.get("out_hop_scale", 1.0)), requiem=opt.get("requiem", "0") == "1")
    label += " [algorithmic bytes for fs %d, fft %d%s]" % (fs, fft, ", Requiem" if opt.get("requiem", "0") == "1" else "")
    print("# HBM traffic per launch from rocprofv3 PMC (separate --pmc passes for FETCH_SIZE and WRITE_SIZE), "
          "%s, %d frames per launch" % (label, frames))
    print("# rocprofv3 reports KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md "
          "HBM section), so")
    print("# corrected_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  algorithmic_MB = DESIGN.md bytes/frame x frames per launch.")
    print("%-24s %12s %12s %14s %16s %8s" % ("kernel", "FETCH_KiB", "WRITE_KiB", "corrected_MB", "algorithmic_MB", "ratio"))
    rows = []
    for k in fetch:
        f, w = fetch[k], write.get(k, 0.0)
        corr = (2 * f + w) * 1024 / 1e6
        rows.append((corr, k, f, w))
    for corr, k, f, w in sorted(rows, reverse=True):
        if k in algo:
            a = algo[k] * frames / 1e6
            print("%-24s %12.0f %12.0f %14.1f %16.1f %8.2f" % (k, f, w, corr, a, corr / a))
        else:
            print("%-24s %12.0f %12.0f %14.1f %16s %8s" % (k, f, w, corr, "-", "-"))


if __name__ == "__main__":
    main()
