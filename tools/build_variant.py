"""Build an experimental variant of librechorus_hip.so for A/B timing on the GPU box without rebuilding there:
    python tools/build_variant.py <tag> <file.hip>[,<file.hip>...] [-DNAME[=V] ...]
recompiles the listed kernel files with the extra defines, links them with the standard objects of the other files
(rechorus_amd/csrc/build/) into tools/bin/lib_<tag>.so; run with RC_LIB_PATH=tools/bin/lib_<tag>.so."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rechorus_amd.csrc import build as B  # noqa: E402


def main():
    tag, files = sys.argv[1], sys.argv[2].split(",")
    defs = sys.argv[3:]
    B.build(verbose=False)
    flags = open(os.path.join(B.OBJ_DIR, ".flags")).read().split()
    out_dir = os.path.join(ROOT, "tools", "bin")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in B.SOURCES:
        std = os.path.join(B.OBJ_DIR, src.replace(".hip", ".o"))
        if src in files:
            o = os.path.join(out_dir, f"{tag}_{src.replace('.hip', '.o')}")
            subprocess.run([B._hipcc()] + flags + defs + ["-c", os.path.join(B.HERE, src), "-o", o], check=True)
            objs.append(o)
        else:
            objs.append(std)
    lib = os.path.join(out_dir, f"lib_{tag}.so")
    subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib] + objs, check=True)
    print("built", lib)


if __name__ == "__main__":
    main()
