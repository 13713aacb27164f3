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
SOURCES = ["api.hip", "store_kvcache.hip", "paged_decode.hip", "prefill_mfma.hip", "rope_store.hip", "layer_ops.hip", "skinny_gemm.hip", "linear_stream.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "nvh_attn.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
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


def _compile(src, asm):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    flags = [*FLAGS, *FILE_FLAGS.get(src, [])]
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


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--asm", action="store_true", help="also emit .s and register/LDS usage per kernel into build/")
    args = ap.parse_args()
    try:
        build(force=args.force, asm=args.asm)
    except subprocess.CalledProcessError as e:
        sys.exit(e.returncode)
