"""Builds wordpiece_amd/libwordpiece_amd.so (HIP kernels + C ABI + C++ API) for gfx950 with hipcc.

In-tree on purpose: the built .so travels to the GPU box with the repo snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwordpiece_amd.so")
LIB_DBG = os.path.join(HERE, "libwordpiece_amd_dbg.so")  # -DWP_DEBUG_BOUNDS (csrc/common.h): checked scatters
RUNNER = os.path.join(HERE, "runner")
CPPTEST = os.path.join(HERE, "test_word_piece")
SOURCES = ["encoder.hip", "word_piece.cpp"]
ARCH = "gfx950"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps():
    out = []
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".hip", ".cpp", ".hpp")):
                out.append(os.path.join(root, f))
    return out


def build(force=False, verbose=False, debug=True):
    hipcc = os.environ.get("HIPCC", "hipcc")
    deps = _deps()
    if force or _newer(LIB, deps):
        cmd = [hipcc, "-O3", "--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function"] + os.environ.get("WP_HIPCC_FLAGS", "").split() + [
               "-x", "hip", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    if debug and (force or _newer(LIB_DBG, deps)):
        cmd = [hipcc, "-O3", "--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
               "-DWP_DEBUG_BOUNDS", "-x", "hip", "-o", LIB_DBG] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    if force or _newer(RUNNER, deps + [LIB]):
        cmd = [hipcc, "-O2", "-std=c++17", "-o", RUNNER, os.path.join(CSRC, "runner.cpp"),
               "-L" + HERE, "-lwordpiece_amd", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    test_src = os.path.join(os.path.dirname(HERE), "tests", "cpp", "test_word_piece.cpp")
    if os.path.exists(test_src) and (force or _newer(CPPTEST, deps + [LIB, test_src])):
        cmd = [hipcc, "-O2", "-std=c++17", "-o", CPPTEST, test_src, "-L" + HERE, "-lwordpiece_amd",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
