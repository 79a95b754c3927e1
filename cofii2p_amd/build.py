"""In-tree build of libcofi_hip.so (hipcc, gfx950 only).  `python -m cofii2p_amd.build`."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcofi_hip.so")
# COFI_HIPCC_FLAGS: extra compiler flags, e.g. -DCOFI_ATTN_ABLATION (the timing-ablation builds of the bf16x6 attention kernel for
# tools/attn_ablate.py); not part of the shipped library
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("COFI_HIPCC_FLAGS", "").split()


# per-file flags.  gemm.hip: the SLP pass packs adjacent scalar fp32 operations of the operand splits into v_pk_* instructions, which are slower
# beside MFMAs (measured neutral-to-negative on the six-product kernels) and hide the v_fma_mix_f32 form of the fp16 split
# (csrc/gemm_f16_big.inc)
FILE_FLAGS = {"gemm.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + [os.path.join(HERE, "..", "include", "cofi_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen([_hipcc()] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj])))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    if verbose:
        print("built", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
