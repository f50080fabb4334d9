"""Build the gfx950 kernel library (``lib/libddl_hip.so``) with hipcc.

The kernels expose a plain C ABI (raw device pointers + a ``hipStream_t``), so no torch headers are needed:
each ``csrc/*.hip`` compiles in seconds, cross-compiles on a box without a GPU, and the resulting ``.so`` is
loaded in-tree with ``ctypes`` (see ``ops/_lib.py``).

    python -m distributed_deep_learning_amd.build [--force] [--jobs N]
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = PKG / "csrc" / "build"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libddl_hip.so"
ARCH = "gfx950"

HIP_FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fno-gpu-rdc",
             "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _newer(src: Path, dst: Path, deps) -> bool:
    if not dst.exists():
        return True
    t = dst.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [src, *deps])


def _compile(hipcc: str, src: Path, obj: Path) -> str:
    cmd = [hipcc, *HIP_FLAGS, "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr}")
    return src.name


def build(force: bool = False, jobs: int | None = None, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(parents=True, exist_ok=True)
    LIBDIR.mkdir(parents=True, exist_ok=True)
    headers = sorted(CSRC.glob("*.h"))
    sources = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.cpp"))
    todo = []
    objs = []
    for s in sources:
        o = OBJ / (s.stem + ".o")
        objs.append(o)
        if force or _newer(s, o, headers):
            todo.append((s, o))
    jobs = jobs or min(8, os.cpu_count() or 1)
    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            for name in ex.map(lambda so: _compile(hipcc, *so), todo):
                if verbose:
                    print(f"[build] compiled {name}", flush=True)
    if force or todo or not LIB.exists():
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB}", flush=True)
    return LIB


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args(argv)
    build(force=a.force, jobs=a.jobs)
    return 0


if __name__ == "__main__":
    sys.exit(main())
