"""Builds iron_amd/csrc/libiron_hip.so (gfx950 only) with hipcc, in-tree.

Used by __graft_entry__.build(); hipcc cross-compiles without a GPU.  The .so is git-ignored but
travels with the tree to the GPU box.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libiron_hip.so")
LIB_TRAIN = os.path.join(CSRC, "libiron_train.so")  # backward passes (include/iron_train.h); hand-written GEMM, no BLAS library
OBJ_DIR = os.path.join(CSRC, "build")
MANIFEST = os.path.join(OBJ_DIR, "manifest.json")  # which flag set every translation unit was compiled with (bench.py echoes it)

SOURCES = ["pack.hip", "pack_h2.hip", "sdf_forward.hip", "h2_kernels.hip", "w16.hip", "pointwise.hip", "trace.hip", "shade.hip", "getall_rev.hip", "envelope.hip", "nerf.hip", "neus.hip", "profile.hip"]
TRAIN_SOURCES = ["train.hip"]
HEADERS = [os.path.join("..", "..", "include", "iron_train.h"), "gemm_h2.h", "lds_dma.h", "iron_common.h", "mlp_core.h", "mlp_h2.h", "mlp_h2_rev.h", "shade_args.h", "h2_setup.h", "pack_common.h", "ggx_core.h", os.path.join("..", "..", "include", "iron_hip.h")]

BASE_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # softplus(beta=100) through v_exp_f32 / v_log_f32 (rel-L2 7.7e-7 vs the fp32 reference on the SDF
    # output; the libm-accurate form, -DIRON_FAST_SOFTPLUS=0, gives 4.0e-7 at half the speed)
    "-DIRON_FAST_SOFTPLUS=1",
    # the pointwise glue must round like the reference's separate torch mul/add kernels
    "-ffp-contract=off",
    "-Wno-comment", "-Wno-unused-result",
]

# Accumulators of the h2 kernels in VGPRs (no v_accvgpr_read in front of the epilogue, 3-5x less scratch).  The pass
# behind this option has crashed hipcc 7.2 on some revisions of trace.hip; a source that fails with it is recompiled
# with weaker flag sets (see compile_one; the kernels are correct either way, only slower).
OPTIONAL_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]

# per-source additions.  getall_rev.hip: its ring step carries a longer staged epilogue than mlp_h2.h's; with the default
# -pragma-unroll-threshold (16 K) LLVM silently leaves the 16-k-step loop of step_hidden_x rolled (runtime stage dispatch, the
# epilogue state in scratch: 19 630 scratch instructions instead of ~200)
SOURCE_FLAGS = {"getall_rev.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _digest(extra_flags) -> str:
    h = hashlib.sha256()
    for name in SOURCES + TRAIN_SOURCES + HEADERS:
        p = os.path.join(CSRC, name)
        if os.path.exists(p):
            with open(p, "rb") as f:
                h.update(f.read())
    h.update(" ".join(BASE_FLAGS + OPTIONAL_FLAGS + list(extra_flags)).encode())
    h.update(json.dumps(SOURCE_FLAGS, sort_keys=True).encode())
    return h.hexdigest()


def manifest() -> dict:
    """The flag set each translation unit of the shipped libraries was compiled with (a unit that hipcc could only build
    with a weaker set is marked `fallback`); {} when the library was built by something else."""
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except OSError:
        return {}


def build(force: bool = False, extra_flags=(), verbose: bool = True) -> str:
    """Compile every .hip source and link the shared library; returns its path."""
    extra_flags = list(extra_flags) + os.environ.get("IRON_HIPCC_FLAGS", "").split()
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, "stamp")
    dig = _digest(extra_flags)
    if (not force and os.path.exists(LIB) and os.path.exists(LIB_TRAIN) and os.path.exists(stamp) and os.path.exists(MANIFEST)
            and open(stamp).read() == dig):
        return LIB
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    used_flags = {}

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        tail = ["-c", os.path.join(CSRC, src), "-o", obj]
        # flag sets tried in order: the full set; the same with the sampler's tile deferral off (the one construct the
        # vgpr-form pass has crashed on); finally without the vgpr-form option
        attempts = [OPTIONAL_FLAGS, OPTIONAL_FLAGS + ["-DIRON_SAMPLER_DEFER=0"], []]
        r = None
        for i, opt in enumerate(attempts):
            per_src = SOURCE_FLAGS.get(src, [])
            r = subprocess.run([hipcc] + BASE_FLAGS + opt + per_src + extra_flags + tail, capture_output=True, text=True)
            if r.returncode == 0:
                used_flags[src] = {"flags": BASE_FLAGS + opt + per_src + extra_flags, "attempt": i, "fallback": i > 0}
                break
            if verbose and i + 1 < len(attempts):
                print("build: %s failed with [%s], retrying with [%s]" % (src, " ".join(opt), " ".join(attempts[i + 1])), file=sys.stderr)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), 6)) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    # the training library: one source; its layer products are the split-fp16 MFMA GEMM of gemm_h2.h (no BLAS library linked)
    cmd = [hipcc] + BASE_FLAGS + extra_flags + ["-shared", "-o", LIB_TRAIN] + [os.path.join(CSRC, s) for s in TRAIN_SOURCES]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for the training library:\n%s\n%s" % (r.stdout, r.stderr))
    for src in TRAIN_SOURCES:
        used_flags[src] = {"flags": BASE_FLAGS + extra_flags, "attempt": 0, "fallback": False}
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip().splitlines()
    with open(MANIFEST, "w") as f:
        json.dump({"hipcc": ver[0] if ver else "?", "digest": dig, "units": used_flags,
                   "any_fallback": any(u["fallback"] for u in used_flags.values())}, f, indent=1, sort_keys=True)
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print("built", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
