"""Hash of the kernel sources: profiles/counters_*.json are stamped with it, and bench.py reports their numbers only while the
sources they were measured on are the ones being run (committed counter files otherwise go stale silently)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "directx-raytracing-spheres-demo_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "directx-raytracing-spheres-demo_amd", "csrc", "*.cpp"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
