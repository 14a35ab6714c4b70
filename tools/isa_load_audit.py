#!/usr/bin/env python3
"""Where does a kernel wait for its global loads?  Compiles one translation unit to gfx950 assembly (hipcc -S, no GPU
needed) and lists, for the kernels whose demangled name contains <filter>, every group of global loads with the number
of instructions up to the first `s_waitcnt vmcnt` behind it and what lies in between (barriers, LDS reads, FP64 work).
A load that is waited for within a few instructions, with no barrier in between, has its whole round trip exposed;
a run of such loads is a chain ("Loads the compiler schedules badly").

usage: tools/isa_load_audit.py <file.hip> <kernel name filter> [--all] [-- extra hipcc flags]
       --all: every load group, not only the ones waited for within 20 instructions"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    show_all = "--all" in args
    args = [a for a in args if a != "--all"]
    src, flt = args[0], args[1] if len(args) > 1 else ""
    asm = "/tmp/isa_load_audit.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
                    "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), src, "-o", asm] + extra, check=True,
                   capture_output=True)
    txt = open(asm).read()
    for m in re.finditer(r"^(_Z\S*):", txt, re.M):
        end = txt.find("s_endpgm", m.start())
        if end < 0:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if flt not in name:
            continue
        ins = [l.split(";")[0].strip() for l in txt[m.start():end].splitlines()
               if l.startswith("\t") and not l.strip().startswith((".", ";")) and l.split(";")[0].strip()]
        print("%s: %d instructions" % (name[:100], len(ins)))
        i = 0
        while i < len(ins):
            if ins[i].startswith(("global_load", "flat_load")):
                j = i
                while j < len(ins) and not ins[j].startswith("s_waitcnt vmcnt"):
                    j += 1
                seg = ins[i:j]
                loads = sum(t.startswith(("global_load", "flat_load")) for t in seg)
                bars = sum(t.startswith("s_barrier") for t in seg)
                lds = sum(t.startswith("ds_read") for t in seg)
                fp = sum(bool(re.match(r"v_(fma|fmac|mul|add)_f64", t)) for t in seg)
                if show_all or (j - i < 20 and bars == 0):
                    print("  %6d: %2d load(s), first wait after %3d instructions (barriers %d, ds_read %d, fp64 %d)  %s"
                          % (i, loads, j - i, bars, lds, fp, ins[i][:60]))
                i = j + 1
            else:
                i += 1


if __name__ == "__main__":
    main()
