"""Experiment builds: python tools/build_variant.py NAME [-DFLAG ...] -> variants/libgsraster_NAME.so
(only binning.hip/composite.hip see the defines; the production library is untouched)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grendel-gs_amd", "csrc"))
import build as B  # noqa: E402


def main():
    name, defs = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(ROOT, "variants")
    os.makedirs(out_dir, exist_ok=True)
    B.build()
    objs = []
    for u in B.UNITS:
        obj = os.path.join(B.HERE, u + ".o")
        if u in ("binning", "composite", "loss") and defs:
            obj = os.path.join(out_dir, f"{u}_{name}.o")
            subprocess.run([B._hipcc()] + B.FLAGS + defs + ["-c", os.path.join(B.HERE, u + ".hip"), "-o", obj], check=True)
        objs.append(obj)
    so = os.path.join(out_dir, f"libgsraster_{name}.so")
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", so], check=True)
    print(so)


if __name__ == "__main__":
    main()
