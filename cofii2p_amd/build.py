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


def _deps(path, seen=None):
    """`path` and every file it includes with #include "..." (csrc/ and include/), transitively."""
    import re

    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as fh:
        for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), flags=re.M):
            for d in (os.path.dirname(path), CSRC, os.path.join(HERE, "..", "include")):
                if os.path.exists(os.path.join(d, name)):
                    _deps(os.path.normpath(os.path.join(d, name)), seen)
                    break
    return seen


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    stamp = " ".join(FLAGS)
    stamp_file = os.path.join(objdir, "flags.txt")
    same_flags = os.path.exists(stamp_file) and open(stamp_file).read() == stamp
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        # an object newer than its source and everything the source includes is kept (same flags, not --force)
        if not force and same_flags and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in _deps(src)):
            continue
        procs.append((src, subprocess.Popen([_hipcc()] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj])))
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    if verbose:
        print("built", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
