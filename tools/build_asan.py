#!/usr/bin/env python3
"""AddressSanitizer build of libworld_hip.so (SURVEY §5 / VERDICT r4 item 8): every translation unit with
`--offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -g`, linked to
python-world_amd/lib/variants/libworld_hip_asan.so.  Run with tools/asan_probe.sh on a GPU box
(HSA_XNACK=1, the ASan runtime preloaded, WH_LIB pointing at the variant)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
import build as B  # noqa: E402

FLAGS = ["--offload-arch=gfx950:xnack+", "-fsanitize=address", "-shared-libsan", "-g", "-O1", "-std=c++17", "-fPIC",
         "-ffp-contract=off", "-Wno-unused-function"]


# Two kernels address a twiddle table by its LDS byte offset and trap unless their dynamic LDS block starts at LDS address 0
# (stonemask_tab_kernel, hv_refine_kernel<TWL>); the sanitizer places LDS bookkeeping of its own in front, so this build
# takes their general forms (the staged StoneMask kernel, refinement twiddles from the global tables).
# The exact phase scan declares 74 KB of static LDS per workgroup (two padded 4096-sample tiles); with the sanitizer's
# LDS redzones the kernel no longer loads ("HSA_STATUS_ERROR_INVALID_ISA" at dispatch): this build scans 1024-sample tiles
# on 128 threads — the same arithmetic (the tile size does not enter the sums).
ASAN_TU_FLAGS = {"wh_stonemask.hip": ["-DWH_STONEMASK_TABLE=0"], "wh_harvest.hip": ["-DWH_HV_LDS_TWIDDLES=0"],
                 "wh_synthesis.hip": ["-DWH_XTILE=1024", "-DWH_XTHREADS=128", "-DWH_PFINISH=256"]}  # (a 1024-thread
# workgroup of the instrumented pulse_finish_kernel does not load either: fewer threads, the same per-pulse arithmetic)


def main():
    odir = os.path.join(B.OBJ_DIR, "variants", "asan")
    os.makedirs(odir, exist_ok=True)
    units = sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip"))

    def one(u):
        obj = os.path.join(odir, u[:-4] + ".o")
        r = subprocess.run([B._hipcc()] + FLAGS + B.TU_FLAGS.get(u, []) + ASAN_TU_FLAGS.get(u, []) + ["-c", os.path.join(B.CSRC, u), "-o", obj], capture_output=True, text=True)
        return u, obj, r.returncode, (r.stdout + r.stderr)[-2000:]

    objs = []
    with ThreadPoolExecutor(max_workers=6) as ex:
        for u, obj, rc, log in ex.map(one, units):
            print(u, "ok" if rc == 0 else "FAILED\n" + log, flush=True)
            if rc != 0:
                return 1
            objs.append(obj)
    vdir = os.path.join(B.OUT_DIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    lib = os.path.join(vdir, "libworld_hip_asan.so")
    r = subprocess.run([B._hipcc(), "--offload-arch=gfx950:xnack+", "-fsanitize=address", "-shared-libsan", "-shared", "-fPIC",
                        "-o", lib] + objs, capture_output=True, text=True)
    print("link", "ok" if r.returncode == 0 else "FAILED\n" + (r.stdout + r.stderr)[-3000:])
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
