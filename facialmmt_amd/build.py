"""Build libfmmt_hip.so (gfx950 only) in-tree with hipcc.  `python -m facialmmt_amd.build`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libfmmt_hip.so")
SOURCES = ["gemm.hip", "layernorm.hip", "attn.hip", "wattn_mfma.hip", "wattn_bwd_ref.hip", "wblock.hip", "wblock_ref.hip", "mha_mfma.hip", "misc.hip", "preproc.hip", "mlp_fused.hip", "mlp_ref.hip", "patch_ln.hip", "lin_lnbwd.hip", "plm_fused.hip", "frame_filter.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, tag: str = "") -> str:
    """`tag` (or FMMT_BUILD_TAG): a second build beside the product library -- objects in build_<tag>/, output libfmmt_hip_<tag>.so --
    for same-call A/B of a development switch (FMMT_CFLAGS=-DFMMT_EXP_...=1; the probes load it through PROBE_LIB)"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = tag or os.environ.get("FMMT_BUILD_TAG", "")
    csrc = os.environ.get("FMMT_CSRC_DIR", CSRC) if tag else CSRC      # a tagged build may come from another source tree (an earlier commit)
    headers = sorted(glob.glob(os.path.join(csrc, "*.h"))) + [os.path.join(csrc, "..", "..", "include", "fmmt.h")]
    objdir = os.path.join(HERE, "build_" + tag if tag else "build")
    out = os.path.join(HERE, f"libfmmt_hip_{tag}.so") if tag else OUT
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src = os.path.join(csrc, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + os.environ.get("FMMT_CFLAGS", "").split() + ["-c", src, "-o", obj]      # FMMT_CFLAGS: development builds
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(out, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
