"""Build librechorus_hip.so (gfx950) in-tree with hipcc.

Usage:  python -m rechorus_amd.csrc.build [--force] [--no-dpp] [--resource-usage]

hipcc cross-compiles for gfx950 without a GPU.  Objects go to rechorus_amd/csrc/build/,
the library to rechorus_amd/librechorus_hip.so (git-ignored, shipped to the GPU box by
gpurun).  Only this target exists: no other --offload-arch, no CPU fall-back.
"""
import argparse
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OBJ_DIR = os.path.join(HERE, "build")
LIB_PATH = os.path.join(PKG, "librechorus_hip.so")
INFO_PATH = os.path.join(PKG, "librechorus_hip.build.json")   # flags / compiler / sha256 of the library as built (bench.py's "build" object)
ARCH = "gfx950"

SOURCES = [
    "library.hip",
    "gather_dot.hip",
    "bpr_loss.hip",
    "bprmf_fused.hip",
    "sort_ids.hip",
    "seg_update.hip",
    "dense_opt.hip",
    "bucket_plan.hip",
    "plan_update.hip",
    "train_step.hip", "small_step.hip",
    "neumf.hip", "neumf_step.hip", "neumf_zhead.hip", "mlp.hip", "tower_tail.hip",
    "sasrec.hip", "sasrec_batch.hip", "seq_layers.hip",
    "listwise_loss.hip",
    "fm_bce.hip",
    "sampler.hip",
    "eval_rank.hip",
    "owner_step.hip",
    "bench_mix.hip",
]
# every header of this directory is a dependency of every object (a hand-kept list went stale once: sas_attn_reg.hpp /
# sas_last_row.hpp were missing, so editing them did not rebuild sasrec_batch.o)
HEADERS = sorted(f for f in os.listdir(HERE) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "rechorus_hip.h")]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm with a gfx950 target)")
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def sha256_of(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _write_info(hipcc, flag_str):
    """what the library was built from, next to it (git-ignored like the .so, travels to the GPU box with it)"""
    import json
    try:
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip().splitlines()
    except Exception as e:
        ver = [repr(e)]
    sha = sha256_of(LIB_PATH)
    try:
        old = json.load(open(INFO_PATH))
        if old.get("so_sha256") == sha and old.get("flags") == flag_str:
            return
    except Exception:
        pass
    with open(INFO_PATH, "w") as f:
        json.dump({"flags": flag_str, "hipcc_version": [l for l in ver if l][:3], "so_sha256": sha, "arch": ARCH,
                   "sources": SOURCES}, f, indent=1)


def build(force=False, no_dpp=False, resource_usage=False, verbose=True, defines=()):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    # -ffp-contract=off: only the fmaf() calls written in the kernels fuse, so every template
    # instantiation of the same expression rounds identically (bit-reproducible across paths)
    flags = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall", "-Wno-unused-function",
             "-ffp-contract=off"]
    if no_dpp:
        flags.append("-DRC_NO_DPP")
    flags += ["-D" + d for d in defines]
    if resource_usage:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    tag = os.path.join(OBJ_DIR, ".flags")
    flag_str = " ".join(flags)
    if not os.path.exists(tag) or open(tag).read() != flag_str:
        force = True
    hdrs = [os.path.join(HERE, h) for h in HEADERS]

    def compile_one(src):
        s = os.path.join(HERE, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if not force and _newer(o, [s] + hdrs):
            return o, None
        cmd = [hipcc] + flags + ["-c", s, "-o", o]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{p.stderr[-8000:]}")
        return o, p.stderr

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    if resource_usage:
        for _, err in results:
            if err:
                sys.stderr.write(err)
    if force or not _newer(LIB_PATH, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"link failed:\n{p.stderr[-8000:]}")
    with open(tag, "w") as f:
        f.write(flag_str)
    _write_info(hipcc, flag_str)
    if verbose:
        print(f"built {LIB_PATH} ({os.path.getsize(LIB_PATH) >> 10} KiB)")
    return LIB_PATH


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--no-dpp", action="store_true", help="debug: DPP reductions via ds_bpermute")
    ap.add_argument("--resource-usage", action="store_true")
    ap.add_argument("--define", action="append", default=[], help="extra -D (experiment switches, e.g. RC_NT)")
    a = ap.parse_args()
    build(force=a.force, no_dpp=a.no_dpp, resource_usage=a.resource_usage, defines=a.define)
