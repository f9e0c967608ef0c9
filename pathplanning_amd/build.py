"""Builds libpphip.so (HIP, gfx950) in-tree: pathplanning_amd/lib/libpphip.so."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libpphip.so")
SOURCES = ["pp_capi.hip", "pp_kernels_basic.hip", "pp_paths.hip", "pp_gvd.hip", "pp_wavefront.hip", "pp_wavefront_tiles.hip", "pp_planner.hip", "pp_rrt.hip", "pp_grid_astar.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "pp_hip.h")]
# -ffp-contract=off: no FMA contraction -- discrete outputs must match the CPU reference bit for bit.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    extra = os.environ.get("PP_EXTRA_HIPCC_FLAGS", "").split()  # tuning experiments, e.g. -DPP_WF_WAVES_PER_SIMD=2
    cmd = [hipcc()] + FLAGS + extra + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


def build_pyplanning(force=False, verbose=True):
    """pybind11 module `pyplanning` (reference's Python surface for the hot path), linked against libpphip.so."""
    import sysconfig
    import pybind11
    build()
    host = os.path.join(HERE, "host")
    ext = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    out = os.path.join(LIB_DIR, "pyplanning" + ext)
    srcs = [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith((".cpp", ".hpp"))] + [os.path.join(HERE, "..", "include", "pp_hip.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(x) <= os.path.getmtime(out) for x in srcs):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
           os.path.join(host, "pyplanning.cpp"), "-o", out, "-L" + LIB_DIR, "-lpphip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


def build_plugin_test(verbose=True):
    """tests/cpp/test_plugin.cpp against the C++ plugin header (run on the GPU box)."""
    build()
    out = os.path.join(LIB_DIR, "test_plugin")
    src = os.path.join(HERE, "..", "tests", "cpp", "test_plugin.cpp")
    cmd = ["g++", "-O2", "-std=c++17", src, "-o", out, "-L" + LIB_DIR, "-lpphip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


def build_row_test(verbose=True):
    """tests/cpp/test_row_primitives.hip: GPU self-test of the DPP row primitives (run on the GPU box)."""
    out = os.path.join(LIB_DIR, "test_row_primitives")
    src = os.path.join(HERE, "..", "tests", "cpp", "test_row_primitives.hip")
    if os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(src), os.path.getmtime(os.path.join(CSRC, "pp_row_primitives.hpp"))):
        return out
    cmd = [hipcc(), "--offload-arch=gfx950", "-O2", "-I", CSRC, "-I", os.path.join(HERE, "..", "include"), src, "-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
    print(build_pyplanning())
    print(build_plugin_test())
    print(build_row_test())
