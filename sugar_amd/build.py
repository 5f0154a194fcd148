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
# Packed-FP32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) are switched off for the whole library:
# on MI355X a wave's packed-FP32 results were occasionally wrong while ANOTHER wave on the same CU issued MFMAs
# (tools/fbank_corun.hip: the front-end kernel against a pure-MFMA burner on a second stream, 62 560 disagreeing
# spectrum bins in 600 launches; 0 with the feature off; 0 against VALU / LDS / memory burners).  Two HIP streams
# (extract_vectors, bench.py's pipelined leg) put exactly that mix on a CU.  DESIGN.md section 5b.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# Exception, checked at every build: the front-end (fbank.hip) writes its complex butterflies as v_pk_* inline asm in the
# operand forms the probe (tools/pkfp32_probe.hip, 1.3e10 executions each beside an MFMA burner) found clean, so the
# assembler must know the instructions: that file is compiled with the feature ON, and _check_packed_forms() then reads
# its ISA and fails the build if ANY packed-FP32 instruction (hand-written or chosen by the compiler) is not in one of
# those forms.  The faulty form is "the low result takes the high half of src1" (op_sel bit of src1 set).
PACKED_FP32_FILES = {"fbank.hip"}
PACKED_FP32_SAFE = {
    "",                                                  # no modifiers
    "neg_lo:[0,1] neg_hi:[0,1]", "neg_hi:[0,1]", "neg_lo:[0,1]",       # negations of src1
    "op_sel:[1,0] op_sel_hi:[0,1]",                      # half swap of src0
    "op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]", "op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]",
    "op_sel:[1,0,0] op_sel_hi:[0,1,1]",                  # FMA, half swap of src0
}
EXTRA_FLAGS = {}
if os.environ.get("SUGAR_ALLOW_PACKED_FP32") is None:
    for _s in SOURCES:
        if _s not in PACKED_FP32_FILES:
            EXTRA_FLAGS[_s] = NO_PACKED_FP32


def _check_packed_forms(hipcc, src):
    """Every v_pk_{add,mul,fma}_f32 in the device ISA of `src` must be in a form of PACKED_FP32_SAFE."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "isa.s")
        subprocess.check_call([hipcc] + FLAGS + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        bad, n = [], 0
        for line in open(out):
            m = re.match(r"\s*v_pk_(add|mul|fma)_f32\s+([^;]*)", line)
            if not m:
                continue
            n += 1
            mods = " ".join(t for t in m.group(2).split() if t.startswith(("op_sel", "neg_", "clamp")))
            if mods not in PACKED_FP32_SAFE:
                bad.append(line.strip())
    if bad:
        raise RuntimeError("%s: packed-FP32 instructions in forms not cleared by tools/pkfp32_probe.hip:\n  %s"
                           % (src, "\n  ".join(sorted(set(bad)))))
    return n


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
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        # the host pass of hipcc does not know the device feature and says so once per file
        out = "\n".join(l for l in out.splitlines() if "packed-fp32-ops' is not a recognized feature" not in l)
        if verbose and out.strip():
            print("[hipcc %s]\n%s" % (src, out))
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    if os.environ.get("SUGAR_ALLOW_PACKED_FP32") is None:
        for src in sorted(PACKED_FP32_FILES):
            n = _check_packed_forms(hipcc, src)
            if verbose:
                print("[packed FP32] %s: %d instructions, all in probe-cleared forms" % (src, n))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
