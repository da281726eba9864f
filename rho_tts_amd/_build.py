"""Build the gfx950 shared library in-tree (``rho_tts_amd/librho_tts_amd.so``).

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "librho_tts_amd.so")
ARCH = "gfx950"
PUBLIC_HEADERS = ("rho_tts_amd.h", "rho_tts_amd_debug.h")     # the drop-in boundary; measurement / test entry points

CXXFLAGS = ["-std=c++17", "-O3", "-fPIC", "-fvisibility=hidden", f"--offload-arch={ARCH}",
            "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
# Translation units whose float results are compared bit-for-bit with the CPU reference
# must not have mul+add fused behind their back.
PER_FILE = {"postprocess.hip": ["-ffp-contract=off"], "encoder.hip": ["-ffp-contract=off"], "features.hip": ["-ffp-contract=off"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X path cannot be built (no CPU fallback exists)")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(os.path.dirname(HERE), "include", h) for h in PUBLIC_HEADERS]
    hdrs.append(os.path.abspath(__file__))
    return max(os.path.getmtime(h) for h in hdrs)


def source_hash() -> str:
    """SHA-256 over the native sources (csrc/*.hip, csrc/*.h, include/*.h, the compiler flags): what identifies a BUILD of
    the library independently of where and when hipcc ran (the .so itself is rebuilt by every fresh checkout and need not come out
    byte-identical).  profiles/*.json record it; bench.py reports counter figures only for the build they were measured on."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files += [os.path.join(os.path.dirname(HERE), "include", h) for h in PUBLIC_HEADERS]
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(CXXFLAGS + [f"{k}:{' '.join(v)}" for k, v in sorted(PER_FILE.items())]).encode())
    return h.hexdigest()


def build_native(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _deps_mtime()
    jobs = []
    for src in sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            jobs.append((s, o, [hipcc, *CXXFLAGS, *PER_FILE.get(src, []), "-c", s, "-o", o]))

    def run(job):
        s, o, cmd = job
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in sources()]
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
