"""Builds the native libraries in-tree (they travel to the GPU box with the snapshot):

  opencalibration_amd/libochip.so   HIP kernels + C ABI (include/ochip.h), gfx950 only
  opencalibration_amd/liboc_host.so C++17 host side mirroring the reference's stage interface

hipcc cross-compiles gfx950 without a GPU, so this runs in the authoring container too.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
             "-Wno-unused-result"]
HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off", "-Wall", "-Wextra",
              "-Wno-unused-parameter"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_ochip(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    out = os.path.join(HERE, "libochip.so")
    if force or _stale(out, deps):
        cmd = [HIPCC, *HIP_FLAGS, "-shared", "-o", out, *srcs]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return out


def build_host(force=False, verbose=False):
    hdir = os.path.join(CSRC, "host")
    srcs = sorted(glob.glob(os.path.join(hdir, "*.cpp")))
    if not srcs:
        return None
    deps = srcs + glob.glob(os.path.join(hdir, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))   # csrc/undistort.hpp is shared with the device side
    out = os.path.join(HERE, "liboc_host.so")
    if force or _stale(out, deps):
        cmd = ["g++", *HOST_FLAGS, "-shared", "-o", out, *srcs, "-I", os.path.join(HERE, "..", "include"),
               "-L", HERE, "-lochip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    return build_ochip(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
