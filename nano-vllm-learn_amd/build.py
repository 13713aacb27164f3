#!/usr/bin/env python3
"""Build libnvh_attn.so (the C-ABI attention library) for gfx950 with plain hipcc.

No torch headers, no pybind: the library is loaded with ctypes (nanovllm_hip/_lib.py).
The .so is written in-tree (nanovllm_hip/lib/) so it travels to the GPU box with the repo snapshot.

    python nano-vllm-learn_amd/build.py [--force] [--asm]
"""
import argparse
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "nanovllm_hip", "lib")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libnvh_attn.so")
SOURCES = ["api.hip", "store_kvcache.hip", "paged_decode.hip", "prefill_mfma.hip", "rope_store.hip", "layer_ops.hip", "skinny_gemm.hip", "linear_stream.hip", "allreduce_oneshot.hip", "qkv_attend.hip"]
HEADERS = ["common.h", "kernels.h", "decode_chunked.h", os.path.join("..", "..", "include", "nvh_attn.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA results stay in the architectural VGPRs.  By default hipcc parks accumulators in the AGPR file
# and pays a v_accvgpr_read/write per element wherever VALU code touches them (softmax on S, rescale of O): 159 such moves in
# the prefill loop.  Measured: prefill +12 % TFLOP/s, decode step -1.3 %, same results (same-box A/B, round 1).
# -amdgpu-kernarg-preload-count=14: the first 14 dwords of a kernel's FLAT leading arguments arrive in user SGPRs with the wave
# (gfx950 kernarg preload) instead of behind an s_load from the kernarg segment: the decode-attention and streaming-GEMM kernels
# put the operands of their first DMA there (measured: attention call 9.22 -> 9.08 us together with the hoisted table loads).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-mllvm", "-amdgpu-kernarg-preload-count=14",
         "-Wall", "-Wno-unused-function", "-Wno-unused-command-line-argument"]
# rope_store.hip must round RoPE's products and sums separately (bit parity with the reference's elementwise fp32 ops);
# HIP's default backend contraction ignores the source pragma, so that file is built with contraction off.
FILE_FLAGS = {"rope_store.hip": ["-ffp-contract=off"], "skinny_gemm.hip": ["-ffp-contract=off"], "linear_stream.hip": ["-ffp-contract=off"]}   # its RoPE epilogue too


def _digest():
    h = hashlib.sha256()
    h.update((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _compile(src, asm, extra=(), obj_dir=None):
    obj = os.path.join(obj_dir or OBJ_DIR, src.replace(".hip", ".o"))
    # variant flags may be scoped to one source: "prefill_mfma.hip:-mllvm" applies -mllvm to that file only
    extra = [e.split(":", 1)[1] if ".hip:" in e else e for e in extra if ".hip:" not in e or e.startswith(src + ":")]
    flags = [*FLAGS, *FILE_FLAGS.get(src, []), *extra]
    cmd = [HIPCC, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
    subprocess.run(cmd, check=True)
    if asm:
        subprocess.run([HIPCC, *flags, "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        os.path.join(CSRC, src), "-o", obj.replace(".o", ".s")], check=False,
                       stderr=open(obj.replace(".o", ".resources.txt"), "w"))
    return obj


def build(force=False, asm=False, verbose=True):
    """Compile every HIP source for gfx950 and link libnvh_attn.so.  Returns the library path."""
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, "digest.txt")
    digest = _digest()
    if not force and not asm and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        if verbose:
            print(f"[nvh build] up to date: {LIB}")
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; set HIPCC")
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda s: _compile(s, asm), SOURCES))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    with open(stamp, "w") as f:
        f.write(digest)
    if verbose:
        print(f"[nvh build] built {LIB}")
    return LIB


def build_variant(out, extra):
    """A/B builds: the same sources with extra compiler flags into another file (select it with NVH_LIB_PATH)."""
    obj_dir = os.path.join(OBJ_DIR, "variant_" + os.path.basename(out))
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda s: _compile(s, False, extra, obj_dir), SOURCES))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
    print(f"[nvh build] built variant {out} with {' '.join(extra)}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--asm", action="store_true", help="also emit .s and register/LDS usage per kernel into build/")
    ap.add_argument("--variant", help="output path of an A/B build with --extra flags (the default library is untouched)")
    ap.add_argument("--extra", nargs="*", default=[], help="extra hipcc flags of the variant, e.g. =-DNVH_DMA_AUX=2 (leading '=' keeps argparse off them)")
    args = ap.parse_args()
    try:
        if args.variant:
            build_variant(args.variant, [e.lstrip("=") for e in args.extra])
            sys.exit(0)
        build(force=args.force, asm=args.asm)
    except subprocess.CalledProcessError as e:
        sys.exit(e.returncode)
