"""Build libganecdotes_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m ganecdotes_amd.build [--force]

The shared object is written in-tree (ganecdotes_amd/libganecdotes_hip.so) so that it
travels with the repository snapshot to the GPU box; it is git-ignored.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libganecdotes_hip.so")
STAMP = LIB + ".srchash"
SOURCES = ["gx_gemm.hip", "gx_swav.hip", "gx_sktab.hip", "gx_conv.hip", "gx_fineq.hip", "gx_simclr.hip", "gx_projhead.hip", "gx_head.hip", "gx_kmeans.hip"]
ARCH = "gfx950"


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")


def _src_hash():
    h = hashlib.sha256()
    # the extra compile flags are part of what the library IS: a build made with other flags must not look up to date
    h.update(("flags:" + " ".join(os.environ.get("GX_EXTRA_FLAGS", "").split())).encode())
    for d in (CSRC, INCLUDE):
        if not os.path.isdir(d):           # a package copied next to a prebuilt .so without the repo's include/
            continue
        for name in sorted(os.listdir(d)):
            if name.endswith((".hip", ".h")):
                with open(os.path.join(d, name), "rb") as f:
                    h.update(name.encode())
                    h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    want = _src_hash()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == want:
                return LIB
    hipcc = _hipcc()
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    cmds = []
    for s in SOURCES:
        o = os.path.join(bdir, s.replace(".hip", ".o"))
        # -fno-slp-vectorize: SLP packs the bf16-split remainders into v_pk_add_f32, which stalls the
        # vector issue port next to MFMAs (bf16x6 GEMM 173 -> 193 TFLOP/s without it)
        cmds.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-fno-slp-vectorize",
                     *os.environ.get("GX_EXTRA_FLAGS", "").split(), "-I", CSRC, "-I", INCLUDE, "-c",
                     os.path.join(CSRC, s), "-o", o])
        objs.append(o)
    from concurrent.futures import ThreadPoolExecutor

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 1)) as ex:    # one hipcc per source file
        list(ex.map(run, cmds))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(want)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
