#!/usr/bin/env python3
"""Static estimate of how often a VALU instruction reads the result of the VALU instruction right before it (gfx950 issues
two adjacent INDEPENDENT VALU instructions of a wave per 4-cycle slot -- profiles/r02_valu_rate.txt -- so such adjacent
dependencies cost a whole slot).   usage: tools/isa_pairs.py <listing.s> <demangled-name filter>"""
import re
import subprocess
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2]
    names = re.findall(r"\.amdhsa_kernel (\S+)", s)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for name, d in zip(names, dem):
        if flt not in d:
            continue
        m = re.search(r"^" + re.escape(name) + r":[^\n]*\n(.*?)^\.Lfunc_end", s, re.S | re.M)
        prev = None  # (dest regs) of the previous instruction if it was VALU, else None
        n_valu = n_dep = n_after_other = 0
        for line in m.group(1).splitlines():
            if not line.startswith("\t") or line.strip().startswith((".", ";")):
                if line and not line.startswith("\t"):
                    prev = None  # label: a branch target
                continue
            body = line.split(";")[0].strip()
            op, _, rest = body.partition(" ")
            ops = [o.strip() for o in rest.split(",")]
            if op.startswith("v_") and not op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_cmp", "v_cmpx")):
                n_valu += 1
                dst = regs(ops[0]) if ops else set()
                src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
                if op.startswith(("v_fmac", "v_mac")):
                    src |= dst
                if prev is None:
                    n_after_other += 1
                elif prev & src:
                    n_dep += 1
                prev = dst
            elif op.startswith("v_cmp"):
                n_valu += 1
                src = set().union(*[regs(o) for o in ops]) if ops else set()
                if prev is None:
                    n_after_other += 1
                elif prev & src:
                    n_dep += 1
                prev = set()
            else:
                prev = None
        short = d[:d.index("(")]
        print(f"valu {n_valu}: reads the previous VALU result {n_dep} ({100 * n_dep / n_valu:.0f} %), follows a non-VALU instruction / label {n_after_other} "
              f"({100 * n_after_other / n_valu:.0f} %), pairable {n_valu - n_dep - n_after_other} ({100 * (n_valu - n_dep - n_after_other) / n_valu:.0f} %) | {short[-100:]}")


if __name__ == "__main__":
    main()
