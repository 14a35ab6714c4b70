#!/usr/bin/env python3
"""Build tuning variants of libworld_hip.so next to the shipped one (python-world_amd/lib/variants/): only the
translation units whose flags differ are recompiled, the rest is linked from python-world_amd/build/*.o.

    tools/build_variants.py name=wh_d4c:-DWH_D4C_MAXR=8,-DWH_D4C_RMAXR=8 other=wh_synthesis:-DWH_RESP_RUN=8 ...

Select one at run time with WH_LIB=python-world_amd/lib/variants/libworld_hip_<name>.so (world/_hip.py)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "python-world_amd")
sys.path.insert(0, PKG)
import build as B  # noqa: E402


def one(spec):
    name, rest = spec.split("=", 1)
    per_tu = {}
    for part in rest.split(";"):
        tu, flags = part.split(":", 1)
        per_tu[tu] = [f for f in flags.split(",") if f]
    odir = os.path.join(B.OBJ_DIR, "variants", name)
    os.makedirs(odir, exist_ok=True)
    objs = []
    for u in sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip")):
        base = u[:-4]
        if base in per_tu:
            obj = os.path.join(odir, base + ".o")
            cmd = [B._hipcc()] + B.FLAGS + B.TU_FLAGS.get(u, []) + per_tu[base] + ["-c", os.path.join(B.CSRC, u), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                return name, False, r.stdout + r.stderr
            objs.append(obj)
        else:
            objs.append(os.path.join(B.OBJ_DIR, base + ".o"))
    vdir = os.path.join(B.OUT_DIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    lib = os.path.join(vdir, "libworld_hip_%s.so" % name)
    r = subprocess.run([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    return name, r.returncode == 0, r.stdout + r.stderr


if __name__ == "__main__":
    B.build(verbose=False)
    with ThreadPoolExecutor(max_workers=6) as ex:
        for name, ok, log in ex.map(one, sys.argv[1:]):
            print(name, "ok" if ok else "FAILED\n" + log[-3000:])
