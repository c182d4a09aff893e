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
# match.hip: the matrix-core matcher folds its accumulators with vector instructions right after the MFMAs; with the
# accumulators in VGPRs (instead of the AGPR half of the file) that needs no v_accvgpr_read per register and tile
HIP_FLAGS_PER_FILE = {"match.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off", "-Wall", "-Wextra",
              "-Wno-unused-parameter"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_ochip(force=False, verbose=False):
    """One object per .hip source (opencalibration_amd/build/, git-ignored), linked into libochip.so: a change to one
    kernel file recompiles that file only.  No relocatable device code: a kernel is launched from the file that defines it."""
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    out = os.path.join(HERE, "libochip.so")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC, *HIP_FLAGS, *HIP_FLAGS_PER_FILE.get(os.path.basename(s), []), "-c", "-o", o, s]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if force or procs or _stale(out, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return out


def build_host(force=False, verbose=False):
    """One object per .cpp (opencalibration_amd/build/host_*.o), compiled in parallel, linked into liboc_host.so."""
    hdir = os.path.join(CSRC, "host")
    srcs = sorted(glob.glob(os.path.join(hdir, "*.cpp")))
    if not srcs:
        return None
    hdrs = glob.glob(os.path.join(hdir, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))   # csrc/undistort.hpp is shared with the device side
    out = os.path.join(HERE, "liboc_host.so")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, "host_" + os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = ["g++", *HOST_FLAGS, "-c", "-o", o, s, "-I", os.path.join(HERE, "..", "include")]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if force or procs or _stale(out, objs):
        cmd = ["g++", "-shared", "-fopenmp", "-o", out, *objs, "-L", HERE, "-lochip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    return build_ochip(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
