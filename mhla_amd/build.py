"""Build libmhla_hip.so for gfx950 in-tree (mhla_amd/lib/), with hipcc.

`python -m mhla_amd.build` or `mhla_amd.build.build()`.  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "capi.hip")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libmhla_hip.so")
import glob  # noqa: E402

# every source / header of the single translation unit: a change to any of them makes the library stale
DEPS = sorted(glob.glob(os.path.join(HERE, "csrc", "*"))) + [os.path.join(os.path.dirname(HERE), "include", "mhla_hip.h")]


# No packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in device code: see DESIGN.md section 5
# (run-to-run differences in the low halves of packed accumulations when workgroups of different roles share a CU).
EXTRA_FLAGS = [] if os.environ.get("MHLA_PACKED_FP32") == "1" else ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++20", "-shared", "-fPIC", "-Wall",
           "-Wno-unused-function"] + EXTRA_FLAGS + [SRC, "-o", LIB + ".tmp"]
    if verbose:
        print("[mhla_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
