"""Build libmbpo_hip.so (gfx950) in-tree with hipcc.  No cmake, no torch extension: the library is a plain
C-ABI shared object (include/mbpo_hip.h) that the Python host loads with ctypes.

    python model-based-policy-optimizers_amd/build.py [--force] [--save-temps]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT_DIR = HERE / "mbpo" / "_lib"
LIB = OUT_DIR / "libmbpo_hip.so"
# objects and -save-temps output stay out of the tree (the tree is what gpurun ships); only the .so lives in-tree
OBJ_DIR = Path(os.environ.get("MBPO_BUILD_DIR", "/tmp/mbpo_hip_build"))

ARCH = "gfx950"
CXXFLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fno-fast-math",          # parity: keep IEEE semantics, precise expf/logf/tanhf
    # SimplifyCFG's common-code sinking merges the identical load sequences of the two weight register sets (chain_run.hpp)
    # into one block addressed through a phi of allocas; the sets then cannot be promoted to registers and live in scratch.
    "-mllvm", "-simplifycfg-sink-common=false",
    "-ffp-contract=on",
    "-Wall",
    "-Wno-unused-function",
]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; this library has no CPU fallback)")


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> Path:
    cc = hipcc()
    OUT_DIR.mkdir(parents=True, exist_ok=True)
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    headers = sorted(CSRC.glob("*.hpp")) + [HERE.parent / "include" / "mbpo_hip.h"]
    srcs = sources()
    jobs = []
    objs = []
    for src in srcs:
        obj = OBJ_DIR / (src.stem + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [cc, *CXXFLAGS, "-c", str(src), "-o", str(obj)]
            if save_temps:
                cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        logs = list(ex.map(run, jobs))
    if save_temps:
        (OBJ_DIR / "resource_usage.txt").write_text("\n".join(logs))
    if jobs or force or _stale(LIB, objs):
        run([cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)])
    return LIB


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(p)
