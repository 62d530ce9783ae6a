#!/usr/bin/env python3
"""Per-kernel register / scratch / instruction-mix summary of a hipcc -save-temps assembly listing.

    cd /tmp/isa && hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fno-fast-math -w -c <repo>/.../pt_kernels.hip -save-temps -o x.o
    python tools/isa_summary.py /tmp/isa/pt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s [filter]
"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        b = m.group(2)
        g = lambda k: int(re.search(r"\.amdhsa_" + k + r" (\d+)", b).group(1))
        meta[m.group(1)] = (g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size"))
    dem = demangle(list(meta))
    # function bodies: from "<name>:" to ".Lfunc_end"
    for name, (vg, sg, scratch) in meta.items():
        d = dem[name]
        if flt and flt not in d:
            continue
        m = re.search(r"^" + re.escape(name) + r":[^\n]*\n(.*?)^\.Lfunc_end", s, re.S | re.M)
        body = m.group(1) if m else ""
        ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = lambda p: sum(1 for i in ins if i.startswith(p))
        short = d[:d.index("(")] if "(" in d else d
        print(f"vgpr {vg:3d} sgpr {sg:3d} scratch {scratch:4d} | valu {c('v_'):5d} salu {c('s_'):5d} ds {c('ds_'):4d} vmem {c('global_') + c('buffer_') + c('flat_') + c('scratch_'):4d} "
              f"smem {c('s_load') + c('s_buffer_load'):3d} readlane {c('v_readlane'):4d} writelane {c('v_writelane'):4d} waitcnt {c('s_waitcnt'):4d} | {short[-110:]}")


if __name__ == "__main__":
    main()
