"""Builds libdbaz_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m dotsboxesaz_amd.build [--force]

The shared library has a plain C ABI (include/dbaz.h) and no torch dependency.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdbaz_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# tree.hip restates numpy float semantics: no FMA contraction there.
UNITS = [
    ("tree.hip", ["-ffp-contract=off"]),
    ("engine.hip", []),
    ("nn.hip", []),
    ("replay.hip", ["-ffp-contract=off"]),  # Kahan-compensated float64 means (pandas group_mean)
    ("train.hip", []),
]
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "dbaz.h"))
    objs, jobs = [], []
    for src, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([HIPCC] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        return r.stdout

    with ThreadPoolExecutor(max_workers=4) as ex:
        outs = list(ex.map(run, jobs))
    if verbose:
        for o in outs:
            if o.strip():
                print(o)
    if jobs or force or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
