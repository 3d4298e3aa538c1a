"""Build libldso_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the repo snapshot."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libldso_hip.so")
SOURCES = ["ba_linearize.hip", "ba_reduce.hip", "ba_solve.hip", "ba_api.hip", "tracker.hip", "images.hip", "trace.hip", "ba_activate.hip", "initializer.hip"]
# -ffp-contract=off: elementwise arithmetic is IEEE and follows the reference's operation order (bit-identical
# energies / residual states); fused multiply-adds are spelled explicitly where they are wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
if os.environ.get("LDSO_STAMPS"):          # device-side phase stamps for scripts/dbg_gn.py (debug builds only, see ba_dev.h)
    FLAGS.append("-DLDSO_STAMPS")


FLAGFILE = os.path.join(HERE, "_obj", "flags.txt")


def flags_changed() -> bool:
    """Objects built with other flags (e.g. a LDSO_STAMPS debug build) must not be reused."""
    try:
        return open(FLAGFILE).read() != " ".join(FLAGS)
    except OSError:
        return True


def needs_build() -> bool:
    if not os.path.exists(OUT) or flags_changed():
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", f) for f in ("ldso_hip.h", "ldso_window.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    objs = []
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    force = force or flags_changed()
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        o = os.path.join(HERE, "_obj", s + ".o")
        objs.append(o)
        if not force and os.path.exists(o) and all(os.path.getmtime(o) > os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h") or f == s) \
                and os.path.getmtime(o) > os.path.getmtime(os.path.join(HERE, "..", "include", "ldso_hip.h")) \
                and os.path.getmtime(o) > os.path.getmtime(os.path.join(HERE, "..", "include", "ldso_window.h")):
            continue
        cmd = ["hipcc", *FLAGS, "-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError(f"hipcc failed on {s}")
        if verbose and out.strip():
            print(out)
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("link failed")
    with open(FLAGFILE, "w") as f:
        f.write(" ".join(FLAGS))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
