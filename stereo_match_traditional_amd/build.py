"""Builds libsmt_hip.so (all csrc/*.hip, gfx950) in-tree with hipcc.  No GPU needed."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libsmt_hip.so")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wno-unused-value"]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(HERE, "host", "*")) + \
        [os.path.join(HERE, "..", "include", "smt.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc] + cflags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, r.stdout, r.stderr))
        return obj

    # one translation unit per kernel family, compiled in parallel, then linked into one library
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    build_host()
    return OUT


def build_host():
    """C++ host mirror (host/smt_host.hpp) + the main.cpp counterparts, plain g++ against the C ABI."""
    exes = []
    for name in ("adcensus_main", "matchers_main", "cblsm_main"):
        exe = os.path.join(HERE, "lib", name)
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", os.path.join(HERE, "host", name + ".cpp"),
               "-o", exe, "-L" + os.path.join(HERE, "lib"), "-lsmt_hip", "-Wl,-rpath,$ORIGIN"]
        if name == "adcensus_main":
            # the batched configuration's exchange goes over RCCL (single process, one communicator per device)
            rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
            cmd += ["-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-L" + os.path.join(rocm, "lib"),
                    "-lrccl", "-lamdhip64", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-Wno-unused-result", "-Wno-deprecated-declarations"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host build failed:\n" + r.stdout + r.stderr)
        exes.append(exe)
    return exes


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
