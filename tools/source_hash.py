"""Hash of the kernel sources: profiles/counters_*.json are stamped with it, and bench.py reports their numbers only while the
sources they were measured on are the ones being run (committed counter files otherwise go stale silently).
Comments and white space do not count: the hash is taken over the token text of csrc/*.h*, csrc/*.cpp and include/*.h, so a
reworded comment does not invalidate a measurement, any change the compiler sees does."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments_and_space(src):
    """C / C++ source without comments and white space (string and character literals are kept verbatim)."""
    out = []
    i, n = 0, len(src)
    while i < n:
        ch = src[i]
        if ch == '"' or ch == "'":
            j = i + 1
            while j < n and src[j] != ch:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1])
            i = j + 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            i = n if j < 0 else j + 2
            if out and not out[-1].isspace():
                out.append(" ")
        elif ch.isspace():
            if out and not out[-1].isspace():
                out.append(" ")
            i += 1
        else:
            out.append(ch)
            i += 1
    return "".join(out)


def kernel_source_hash():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "directx-raytracing-spheres-demo_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "directx-raytracing-spheres-demo_amd", "csrc", "*.cpp"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(strip_comments_and_space(open(f, encoding="utf-8").read()).encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())


def kernel_knobs():
    """the PT_* environment knobs that change what the kernels do (read once, at pt_create): part of the counters' stamp -- counters taken
    under one set of knobs say nothing about a run under another"""
    import os
    skip = {"PT_HIP_LIB", "PT_BENCH_REHEARSAL", "PT_PROFILE_STEPS", "PT_PROFILE_WARMUP", "PT_PROFILE_PMC_STEPS", "PT_PROFILE_PINNED", "PT_ROCTX"}
    # PT_PROFILE_PINNED (profiles/collect.sh): knobs set only to make the serialised counter passes run the schedule that the throughput run
    # chooses by itself (a frame submitted to an idle context would otherwise take the latency form) -- not a different configuration
    skip |= set(os.environ.get("PT_PROFILE_PINNED", "").split(","))
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("PT_") and k not in skip}
