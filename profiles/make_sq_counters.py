"""Summarises a rocprofv3 SQ-counter pass of bench.py (--kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU, one frame in flight) per bounce_kernel class.

usage: python profiles/make_sq_counters.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def main():
    path, out_path = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "bounce_kernel<" not in n:
            continue
        args = n[n.index("bounce_kernel<") + len("bounce_kernel<"):].split(">")[0].split(", ")  # kLds, StackT, kPrimary, kLoop, kMulti, kTex
        key = "bounce<loop>" if args[3] == "true" else ("bounce<primary>" if args[2] == "true" else "bounce<compact>")
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            launches[key] += 1
    out = {"source": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                     "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -- python3 bench.py --steps 20 --warmup 2 --frames-in-flight 1 (profiles/collect.sh)",
           "kernels": {}}
    for k, c in agg.items():
        n, waves, wc = launches[k], c["SQ_WAVES"], c["SQ_WAVE_CYCLES"]
        out["kernels"][k] = {
            "launches": n, "waves_per_launch": waves / n, "valu_insts_per_wave": c["SQ_INSTS_VALU"] / waves,
            "valu_insts_per_launch": c["SQ_INSTS_VALU"] / n,
            # time a launch would take if the VALUs never idled: 4 cycles per wave64 instruction on each of 256 CUs x 4 SIMDs at 2.35 GHz
            "valu_bound_us_per_launch": c["SQ_INSTS_VALU"] / n * 4 / 1024 / 2.35e3,
            "active_frac": c["SQ_ACTIVE_INST_ANY"] / wc, "wait_any_frac": c["SQ_WAIT_ANY"] / wc, "wait_inst_frac": c["SQ_WAIT_INST_ANY"] / wc,
            "valu_active_frac_of_wave": c["SQ_ACTIVE_INST_VALU"] / wc,
        }
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
