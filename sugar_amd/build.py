"""
Build libsugar_amd.so (hipcc, gfx950 only) in-tree next to this file.

    python -m sugar_amd.build [--force]

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU box with the tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["conv_gemm.hip", "conv1x1.hip", "res2.hip","fbank.hip", "se_pool.hip", "pool.hip", "score.hip", "bn_recal.hip", "asnorm.hip", "subnet.hip"]
HEADERS = ["common.h", os.path.join("..", "..", "include", "sugar_amd.h")]
LIB = os.path.join(HERE, "libsugar_amd.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=on"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(CSRC, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if verbose and out.strip():
            print("[hipcc %s]\n%s" % (src, out))
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
