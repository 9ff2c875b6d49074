"""Experiment builds: python tools/build_variant.py NAME [--src unit=path.hip ...] [-DFLAG ...] -> variants/libgsraster_NAME.so
-D flags are seen by binning / composite / loss; --src replaces one translation unit by another source file (e.g. an
older revision: git show REV:grendel-gs_amd/csrc/composite.hip > variants/composite_old.hip).  The production library
is untouched."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grendel-gs_amd", "csrc"))
import build as B  # noqa: E402


def main():
    name, rest = sys.argv[1], sys.argv[2:]
    srcs, defs = {}, []
    i = 0
    while i < len(rest):
        if rest[i] == "--src":
            u, path = rest[i + 1].split("=", 1)
            srcs[u] = os.path.abspath(path)
            i += 2
        else:
            defs.append(rest[i])
            i += 1
    out_dir = os.path.join(ROOT, "variants")
    os.makedirs(out_dir, exist_ok=True)
    B.build()
    objs = []
    for u in B.UNITS:
        obj = os.path.join(B.HERE, u + ".o")
        if u in srcs or (u in ("binning", "composite", "loss", "optim", "preprocess", "exchange") and defs):
            obj = os.path.join(out_dir, f"{u}_{name}.o")
            src = srcs.get(u, os.path.join(B.HERE, u + ".hip"))
            subprocess.run([B._hipcc()] + B.FLAGS + defs + ["-I", B.HERE, "-c", src, "-o", obj], check=True)
        objs.append(obj)
    so = os.path.join(out_dir, f"libgsraster_{name}.so")
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", so], check=True)
    print(so)


if __name__ == "__main__":
    main()
