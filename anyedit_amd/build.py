"""Builds anyedit_amd/libanyedit_hip.so (gfx950 code object) IN-TREE with hipcc.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libanyedit_hip.so")
SOURCES = ["c_api.hip", "gemm_conv.hip", "gemm_rowpanel.hip", "ff_fused.hip", "xattn_fused.hip", "attention.hip", "attention_fast.hip", "attention_fp8.hip", "attention_bwd.hip", "norm.hip", "elementwise.hip", "backward.hip", "gate.hip", "expert_kv.hip", "msda.hip",
           "sam_decoder.hip"]
# Per-file flags.
#  * -ffinite-math-only (attention kernels): no NaN / Inf semantics are relied on (masked logits are a finite -1e30) -> fmaxf compiles to
#    bare v_max / v_max3.
#  * -fno-slp-vectorize (attention_fast, gemm_conv, gemm_rowpanel): hipcc's SLP pass packs adjacent fp32 multiplies / adds (softmax rescale
#    and row sums; the epilogues' bias / activation / statistics arithmetic: 47 000 packed ops in gemm_conv.s) into v_pk_mul_f32 /
#    v_pk_add_f32, which cost more than two plain VALU ops beside MFMAs (MI355X_MICROARCH.md, "price of one filler").  Outputs bit-identical;
#    self-attention +1.1 % (d = 40), +1.5 % (d = 80), SAM global attention +2.2 % (profiles/r03_v49_attn_no_slp.txt); UNet step
#    13.50 -> 13.44 ms in alternating A/B runs with the two GEMM files (profiles/r03_v50_gemm_no_slp.txt).  Not applied (not measured) to
#    attention.hip / attention_bwd.hip; norm.hip has no MFMA to compete with.  `python -m anyedit_amd.build --variant` builds an A/B copy.
EXTRA = {"gemm_conv.hip": ["-fno-slp-vectorize"], "gemm_rowpanel.hip": ["-fno-slp-vectorize"], "ff_fused.hip": ["-fno-slp-vectorize"], "xattn_fused.hip": ["-fno-slp-vectorize"], "attention.hip": ["-ffinite-math-only"],
         "attention_fast.hip": ["-ffinite-math-only", "-fno-slp-vectorize"], "attention_fp8.hip": ["-ffinite-math-only"],
         "attention_bwd.hip": ["-ffinite-math-only"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + f.read())
    h.update((" ".join(FLAGS) + repr(sorted(EXTRA.items()))).encode())
    return h.hexdigest()


def build_library(force=False, verbose=True, variant=None, extra_flags=None):
    """Builds the library.  `variant` (a tag) + `extra_flags` ({"file.hip": [flags]}, appended to EXTRA's) builds a second copy
    `libanyedit_hip_<tag>.so` beside the product library for an A/B of compile flags on one GPU box (`AE_LIB_PATH` selects it at import)."""
    lib = LIB if variant is None else os.path.join(HERE, f"libanyedit_hip_{variant}.so")
    extra = {k: list(v) for k, v in EXTRA.items()}
    for k, v in (extra_flags or {}).items():
        if k not in SOURCES:
            raise ValueError(f"{k} is not a source of the library")
        extra[k] = extra.get(k, []) + list(v)
    stamp = lib + ".sha256"
    dig = _digest() if variant is None else None
    if not force and dig is not None and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return lib
    objdir = os.path.join(HERE, "build" if variant is None else f"build_{variant}")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + extra.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    if dig is not None:
        with open(stamp, "w") as f:
            f.write(dig)
    if verbose:
        print(f"built {lib}")
    return lib


if __name__ == "__main__":
    # python -m anyedit_amd.build [--force]
    # python -m anyedit_amd.build --variant noslp attention_bwd.hip=-fno-slp-vectorize attention.hip=-fno-slp-vectorize
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        flags = {}
        for spec in sys.argv[i + 2:]:
            f, _, fl = spec.partition("=")
            flags.setdefault(f, []).append(fl)
        build_library(force=True, variant=sys.argv[i + 1], extra_flags=flags)
    else:
        build_library(force="--force" in sys.argv)
