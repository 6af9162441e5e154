"""Builds libdbaz_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m dotsboxesaz_amd.build [--force]

The shared library has a plain C ABI (include/dbaz.h) and no torch dependency.
"""
import hashlib
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
    ("train_net.hip", []),
]
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def source_files():
    fs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    return fs + [os.path.join(os.path.dirname(HERE), "include", "dbaz.h")]


def source_hash(only=None):
    """sha256 (16 hex digits) over the kernel sources and headers: what dbaz_build_info() of a library built from them reports
    as src=...; only=("nn.hip", "nn.h", "common.h") gives the nn=... hash of the network kernels alone."""
    h = hashlib.sha256()
    for f in source_files():
        if only is not None and os.path.basename(f) not in only:
            continue
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


NN_SOURCES = ("nn.hip", "nn.h", "common.h")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, debug=False):
    """debug=True: libdbaz_hip_debug.so with -DDBAZ_DEBUG (A/B tilings nn_precision 2/3/4, DBAZ_TRAIN_WGRAD_F32); objects in
    csrc/_debug/.  Load it with DBAZ_LIB=<path> (dotsboxesaz_amd/_lib.py)."""
    if debug:
        return _build(force, verbose, ["-DDBAZ_DEBUG"], os.path.join(CSRC, "_debug"), os.path.join(HERE, "libdbaz_hip_debug.so"))
    return _build(force, verbose, [], CSRC, LIB)


def _build(force, verbose, defs, objdir, lib):
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "dbaz.h"))
    objs, jobs = [], []
    for src, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([HIPCC] + COMMON + defs + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        return r.stdout

    # build identity (csrc/buildinfo.cpp): recompiled whenever the hash of the sources changes
    sh = source_hash() + " nn=" + source_hash(NN_SOURCES)
    bo, bh = os.path.join(objdir, "buildinfo.o"), os.path.join(objdir, "buildinfo.hash")
    objs.append(bo)
    if force or not os.path.exists(bo) or not os.path.exists(bh) or open(bh).read() != sh:
        jobs.append(["g++", "-O1", "-fPIC", "-c", os.path.join(CSRC, "buildinfo.cpp"), "-o", bo, '-DDBAZ_SRC_HASH="%s"' % sh] + defs)

    with ThreadPoolExecutor(max_workers=4) as ex:
        outs = list(ex.map(run, jobs))
    open(bh, "w").write(sh)
    if verbose:
        for o in outs:
            if o.strip():
                print(o)
    if jobs or force or _newer(lib, objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, debug="--debug" in sys.argv))
