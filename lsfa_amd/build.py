"""In-tree build of liblsfa_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m lsfa_amd.build [--force]

Flags that matter for parity: -ffp-contract=off (fused multiply-adds only where the
source says fmaf — see DESIGN.md, "Floating-point contract"); division and sqrt stay
correctly rounded (hipcc default), no fast-math.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "liblsfa_hip.so")
SOURCES = ["runtime.hip", "warp.hip", "aggregate.hip", "psroi.hip", "nms.hip", "proposal.hip", "detpost.hip", "dcn.hip", "mv.hip", "conv.hip", "stem.hip", "flownet.hip", "rpn_head.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-I", CSRC] + os.environ.get("LSFA_HIPCC_EXTRA", "-fno-slp-vectorize -fno-vectorize").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
              [os.path.join(INCLUDE, "lsfa_hip.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for msg in ex.map(cc, jobs):
            if verbose and msg:
                print(msg)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr)
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
