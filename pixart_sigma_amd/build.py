"""Builds pixart_sigma_amd/libpixart_hip.so (gfx950 only) from csrc/*.hip with hipcc, in-tree.

    python -m pixart_sigma_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is git-ignored but travels to the
GPU box with the repo snapshot.  Objects are cached per source hash under pixart_sigma_amd/csrc/.obj/.
"""
import argparse
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, ".obj")
LIB = os.path.join(HERE, "libpixart_hip.so")
# (library file, extra compile flags): bf16 operands (training / default) and IEEE-fp16 operands (the 1e-3 forward-parity build)
VARIANTS = {"bf16": ("libpixart_hip.so", []), "f16": ("libpixart_hip_f16.so", ["-DPXA_OPERAND_F16"])}
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# Per-source flags.  attn.hip: the SLP vectoriser packs adjacent scalar fp32 multiplies / adds of the softmax into v_pk_* instructions, which cost more
# issue time beside MFMAs than the scalar forms they replace (MI355X_MICROARCH.md, price of fillers) and re-pair values against the bf16 packing:
# without it forward -1.3 %, dK/dV -1.9 %, dQ (with delta folded into dP) -3.6 % (profiles/r02n_attn_fold_ab.txt).
PER_FILE_FLAGS = {"attn.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(path, extra=()):
    h = hashlib.sha256()
    for p in [path, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_params.h"), os.path.join(INCLUDE, "pixart_hip.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join([*FLAGS, *PER_FILE_FLAGS.get(os.path.basename(path), []), *extra]).encode())
    return h.hexdigest()[:16]


def build(force=False, verbose=False, variants=("bf16", "f16")):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hipcc = _hipcc()
    jobs, plan, keep = [], [], set()
    for v in variants:
        libname, extra = VARIANTS[v]
        objs = []
        for s in srcs:
            src = os.path.join(CSRC, s)
            obj = os.path.join(OBJ, f"{s[:-4]}.{v}.{_digest(src, extra)}.o")
            objs.append(obj)
            keep.add(obj)
            if force or not os.path.exists(obj):
                jobs.append((src, obj, extra))
        plan.append((os.path.join(HERE, libname), objs))

    def compile_one(job):
        src, obj, extra = job
        cmd = [hipcc, *FLAGS, *PER_FILE_FLAGS.get(os.path.basename(src), []), *extra, "-I", INCLUDE, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    for lib, objs in plan:
        stale = not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs)
        if stale or force:
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    # drop objects of older source versions (only when every variant was planned, so a partial build keeps the other's cache)
    if set(variants) == set(VARIANTS):
        for f in os.listdir(OBJ):
            p = os.path.join(OBJ, f)
            if p not in keep:
                os.remove(p)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.verbose))
    sys.exit(0)
