"""Build libmhla_hip.so for gfx950 in-tree (mhla_amd/lib/), with hipcc.

`python -m mhla_amd.build` or `mhla_amd.build.build()`.  hipcc cross-compiles without a GPU.  The library consists of several
translation units (csrc/capi*.hip) compiled side by side and linked; objects are cached under lib/obj/ and rebuilt when any
source, header or the flag set changed.
"""
import glob
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "libmhla_hip.so")
SOURCES = sorted(glob.glob(os.path.join(CSRC, "capi*.hip")))
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(os.path.dirname(HERE), "include", "mhla_hip.h")]

# No packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in device code: see DESIGN.md section 5
# (run-to-run differences in the low halves of packed accumulations when workgroups of different roles share a CU).
PACKED_FP32 = os.environ.get("MHLA_PACKED_FP32") == "1"
DEVICE_FLAGS = [] if PACKED_FP32 else ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# what the library reports through mhla_build_flags(); _lib.load() refuses a library built otherwise
FLAGS_TAG = "gfx950 -O3 " + ("packed-fp32" if PACKED_FP32 else "no-packed-fp32")
EXTRA_DEFINES = [d for d in os.environ.get("MHLA_BUILD_DEFINES", "").split() if d]   # tools/build_variant.sh: -DFOO ablation switches


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _stamp() -> str:
    """Hash of everything an object depends on: headers, flags (a header change rebuilds every unit)."""
    h = hashlib.sha256()
    for fn in HEADERS:
        h.update(open(fn, "rb").read())
    h.update(repr((DEVICE_FLAGS, FLAGS_TAG, EXTRA_DEFINES)).encode())
    return h.hexdigest()[:16]


def _obj_for(src: str, obj_dir: str) -> str:
    return os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")


def _unit_stale(src: str, obj: str, stamp: str) -> bool:
    tag = obj + ".stamp"
    if not (os.path.exists(obj) and os.path.exists(tag)):
        return True
    want = stamp + ":" + hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    return open(tag).read().strip() != want


def _lib_stamp(stamp: str) -> str:
    h = hashlib.sha256(stamp.encode())
    for src in SOURCES:
        h.update(open(src, "rb").read())
    return h.hexdigest()[:16]


def is_stale(lib: str = LIB) -> bool:
    """The library carries the hash of its sources, headers and flags beside it (the object cache need not travel with it)."""
    tag = lib + ".stamp"
    return not (os.path.exists(lib) and os.path.exists(tag) and open(tag).read().strip() == _lib_stamp(_stamp()))


def build(force: bool = False, verbose: bool = True, lib: str = LIB, obj_dir: str = OBJ_DIR) -> str:
    if not force and not is_stale(lib):
        return lib
    stamp = _stamp()
    todo = [s for s in SOURCES if force or _unit_stale(s, _obj_for(s, obj_dir), stamp)]
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = hipcc_path()
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-function",
            f'-DMHLA_BUILD_FLAGS="{FLAGS_TAG}"'] + EXTRA_DEFINES + DEVICE_FLAGS

    def compile_unit(src: str) -> None:
        obj = _obj_for(src, obj_dir)
        cmd = base + ["-c", src, "-o", obj + ".tmp"]
        if verbose:
            print("[mhla_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        # the x86 host pass does not know the gfx950 target feature and says so once per unit: not an error
        err = "\n".join(ln for ln in r.stderr.splitlines() if "is not a recognized feature for this target" not in ln)
        if err.strip() and verbose:
            print(err, file=sys.stderr, flush=True)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd, r.stdout, r.stderr)
        os.replace(obj + ".tmp", obj)
        with open(obj + ".stamp", "w") as f:
            f.write(stamp + ":" + hashlib.sha256(open(src, "rb").read()).hexdigest()[:16])

    if todo:
        with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
            list(ex.map(compile_unit, todo))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj_for(s, obj_dir) for s in SOURCES] + ["-o", lib + ".tmp"]
    if verbose:
        print("[mhla_amd.build]", " ".join(link), flush=True)
    subprocess.run(link, check=True)
    os.replace(lib + ".tmp", lib)
    with open(lib + ".stamp", "w") as f:
        f.write(_lib_stamp(stamp))
    return lib


# ---- the C++ autograd nodes (csrc_torch/mhla_torch.cpp): a host-only library against libtorch, no device code ----
TORCH_SRC = os.path.join(HERE, "csrc_torch", "mhla_torch.cpp")
TORCH_LIB = os.path.join(LIB_DIR, "libmhla_torch.so")


def _torch_stamp() -> str:
    import torch
    h = hashlib.sha256(open(TORCH_SRC, "rb").read())
    h.update(open(os.path.join(os.path.dirname(HERE), "include", "mhla_hip.h"), "rb").read())
    h.update(torch.__version__.encode())
    return h.hexdigest()[:16]


def torch_ext_stale() -> bool:
    tag = TORCH_LIB + ".stamp"
    return not (os.path.exists(TORCH_LIB) and os.path.exists(tag) and open(tag).read().strip() == _torch_stamp())


def build_torch_ext(force: bool = False, verbose: bool = True) -> str:
    """g++ -shared against the installed libtorch (ROCm build): mhla_amd/lib/libmhla_torch.so.  Loaded with
    torch.ops.load_library by mhla_amd/_native.py; binds libmhla_hip.so's C ABI with dlopen at run time."""
    if not force and not torch_ext_stale():
        return TORCH_LIB
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths("cuda") + ["/opt/rocm/include"]
    libs = ce.library_paths("cuda")
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"] + [f"-I{i}" for i in inc] +
           [TORCH_SRC, "-o", TORCH_LIB + ".tmp"] + [f"-L{d}" for d in libs] + ["-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-ldl"] +
           [f"-Wl,-rpath,{d}" for d in libs])
    if verbose:
        print("[mhla_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(TORCH_LIB + ".tmp", TORCH_LIB)
    with open(TORCH_LIB + ".stamp", "w") as f:
        f.write(_torch_stamp())
    return TORCH_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
    build_torch_ext(force="--force" in sys.argv)
    print(TORCH_LIB)
