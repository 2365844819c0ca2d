"""Builds the native pieces in-tree (so they travel to the GPU box with the snapshot).

* ``pgen_rs_amd/libpgen_hip.so`` — HIP kernels + C ABI, ``hipcc --offload-arch=gfx950``.

(The CPU oracle under ``oracle/`` is test infrastructure and is built by
``__graft_entry__.build()`` / its own Makefile, not from here.)

hipcc cross-compiles for gfx950 without a GPU present.  Re-builds only when a source is newer
than its target.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
HIP_LIB = PKG_DIR / "libpgen_hip.so"

HIP_SOURCES = ["capi.hip", "host_pure.cpp", "gt_rows.hip", "gt_flat.hip", "gt_wide.hip", "gt_scan.hip", "gt_pick.hip", "gt_rowpick.hip"]
HIPCC_FLAGS = [
    "-O3",
    "-std=c++17",
    "--offload-arch=gfx950",
    "-fPIC",
    "-shared",
    "-Wall",
    "-Wextra",
    "-Wno-unused-parameter",
    "-fno-gpu-rdc",
    # the work-queue claims are single-lane returning atomics whose result is consumed a whole step later;
    # the wave-reduction rewrite of the atomic optimizer would read the result back at once (s_waitcnt vmcnt(0))
    "-mllvm",
    "-amdgpu-atomic-optimizer-strategy=None",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libpgen_hip.so)")


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def _run(cmd: list[str], cwd: Path) -> None:
    proc = subprocess.run(cmd, cwd=str(cwd), capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout)
        sys.stderr.write(proc.stderr)
        raise RuntimeError(f"build command failed ({proc.returncode}): {' '.join(cmd)}")
    if proc.stderr.strip():
        sys.stderr.write(proc.stderr)


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    srcs = [CSRC / s for s in HIP_SOURCES]
    deps = srcs + sorted(CSRC.glob("*.h")) + [REPO_ROOT / "include" / "pgen_hip.h", Path(__file__)]
    if force or _stale(HIP_LIB, deps):
        cmd = [_hipcc(), *HIPCC_FLAGS, "-I", str(REPO_ROOT / "include"), "-o", str(HIP_LIB), *map(str, srcs)]
        if verbose:
            print(" ".join(cmd))
        _run(cmd, CSRC)
    return HIP_LIB


HOST_DIR = PKG_DIR / "host"
HOST_BIN = PKG_DIR / "pgen-hip"
HOST_SOURCES = ["cli.cpp", "pfile.cpp", "csvlite.cpp", "expr.cpp", "bgzf.cpp"]


def build_host(force: bool = False, verbose: bool = False) -> Path:
    """The C++ host (pgen-rs's Pfile/CLI surface) — plain g++, links only the C ABI."""
    srcs = [HOST_DIR / s for s in HOST_SOURCES]
    deps = srcs + sorted(HOST_DIR.glob("*.h")) + [REPO_ROOT / "include" / "pgen_hip.h", HIP_LIB, Path(__file__)]
    if force or _stale(HOST_BIN, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-pthread", "-I", str(REPO_ROOT / "include"),
               "-o", str(HOST_BIN), *map(str, srcs), "-L", str(PKG_DIR), "-lpgen_hip", "-lz", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        _run(cmd, HOST_DIR)
    return HOST_BIN


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip(force=force, verbose=verbose)
    build_host(force=force, verbose=verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("built", HIP_LIB, "and", HOST_BIN)
