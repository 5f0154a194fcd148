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
SOURCES = ["conv_gemm.hip", "conv1x1.hip", "res2.hip","fbank.hip", "se_pool.hip", "pool.hip", "score.hip", "bn_recal.hip", "asnorm.hip", "subnet.hip", "hostpack.hip", "arena.hip"]
HEADERS = ["common.h", os.path.join("..", "..", "include", "sugar_amd.h")]
LIB = os.path.join(HERE, "libsugar_amd.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=on"]
# Packed-FP32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): on MI355X a wave's packed-FP32 results were
# occasionally wrong while ANOTHER wave on the same CU issued MFMAs (tools/fbank_corun.hip: the front-end kernel against a
# pure-MFMA burner on a second stream, 62 560 disagreeing spectrum bins in 600 launches; 0 against VALU / LDS / memory
# burners).  Two HIP streams (extract_vectors, bench.py's pipelined leg) put exactly that mix on a CU.  DESIGN.md 5b.
# tools/pkfp32_probe.hip isolated the fault to ONE operand form: the op_sel bit of src1 set (the low result takes the
# HIGH half of src1; 1 wrong result in 13 000 beside an MFMA burner, always lanes 48-63).  The feature stays OFF for the
# library, with one exception: the front-end (fbank.hip) writes its complex butterflies as v_pk_* inline asm (the kernel
# was VALU-bound and complex arithmetic is two-lane by nature), so the assembler must know the instructions and that file
# is compiled with the feature on.  Every build then reads the device ISA of such a file and refuses any packed-FP32
# instruction -- hand-written or picked by hipcc -- whose form is not in PACKED_FP32_SAFE: forms that ran 1.3e10 times
# each beside the burner with 0 wrong results (profiles/r02_pkfp32_probe_forms.log; the faulty form is the log's
# positive control).  SUGAR_PACKED_FP32_ALL=1 compiles every source that way (all 2 136 instructions hipcc then picks
# are in cleared forms; measured in one job: 1.248 ms per step either way, so it is not the default).
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
PACKED_FP32_SAFE = {
    "",                                                  # no modifiers
    "neg_lo:[0,1] neg_hi:[0,1]", "neg_hi:[0,1]", "neg_lo:[0,1]",       # negations of src1
    "op_sel_hi:[1,0]", "op_sel_hi:[0,1]",                # low half of src1 / src0 for both results
    "op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]",
    "op_sel_hi:[1,0,1]", "op_sel_hi:[0,1,1]", "op_sel_hi:[1,1,0]",     # FMA, low-half broadcasts
    "op_sel:[1,0,0]",                                    # FMA, high half of src0 for both results
    "op_sel:[1,0] op_sel_hi:[0,1]",                      # half swap of src0
    "op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]", "op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]",
    "op_sel:[1,0,0] op_sel_hi:[0,1,1]",                  # FMA, half swap of src0
}
PACKED_FP32_FILES = set(SOURCES) if os.environ.get("SUGAR_PACKED_FP32_ALL") else {"fbank.hip"}
EXTRA_FLAGS = {_s: NO_PACKED_FP32 for _s in SOURCES if _s not in PACKED_FP32_FILES}


def _check_packed_forms(hipcc, src):
    """Every v_pk_{add,mul,fma}_f32 in the device ISA of `src` must be in a form of PACKED_FP32_SAFE."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "isa.s")
        subprocess.check_call([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out],
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
    checks = [(src, subprocess.Popen([sys.executable, "-c",
                                      "import sys; sys.path.insert(0, %r); from sugar_amd import build; "
                                      "print(build._check_packed_forms(%r, %r))" % (os.path.dirname(HERE), hipcc, src)],
                                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
              for src in sorted(PACKED_FP32_FILES)]
    total = 0
    for src, p in checks:
        out, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(err.strip().splitlines()[-1] if err.strip() else "packed-FP32 check failed for " + src)
        total += int(out.strip())
    if verbose:
        print("[packed FP32] %d instructions in %d files, all in probe-cleared forms" % (total, len(checks)))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
