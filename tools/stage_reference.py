"""Stage the reference's Python next to the repo for ONE gpurun session (scratch, never committed).

The GPU box has no /root/reference.  The level-B1 graft tests (tests/test_gpu_reference_b1.py) execute the
REFERENCE's own gaussian_renderer / scene / arguments / utils / train.py on this repo's HIP operator module, so
the tree has to travel with the gpurun snapshot: this script copies its .py files (1 MB; no submodules, no
examples, no blobs) to `_refstage/reference/`, which .gitignore lists (history stays free of reference sources)
and .gpurunignore does not.  `python tools/stage_reference.py --clean` removes it again.
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference"
DST = os.path.join(ROOT, "_refstage", "reference")


def main():
    if "--clean" in sys.argv:
        shutil.rmtree(os.path.join(ROOT, "_refstage"), ignore_errors=True)
        print("removed", os.path.join(ROOT, "_refstage"))
        return
    if not os.path.isdir(os.path.join(SRC, "gaussian_renderer")):
        raise SystemExit(f"{SRC} is not present: nothing to stage")
    shutil.rmtree(DST, ignore_errors=True)
    n = 0
    for d, dirs, files in os.walk(SRC):
        dirs[:] = [x for x in dirs if x not in (".git", "__pycache__", "submodules", "examples")]
        for f in files:
            if not f.endswith(".py"):
                continue
            rel = os.path.relpath(os.path.join(d, f), SRC)
            os.makedirs(os.path.dirname(os.path.join(DST, rel)), exist_ok=True)
            shutil.copyfile(os.path.join(d, f), os.path.join(DST, rel))
            n += 1
    print(f"staged {n} files -> {DST}")


if __name__ == "__main__":
    main()
