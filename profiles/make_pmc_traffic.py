"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of `bench.py` into
profiles/pmc_traffic.json: HBM bytes per launch for the kernel classes bench.py's roofline object reports.

MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports exactly half of the
bytes of a wide coalesced (16 B/lane) streaming read, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.

usage: python profiles/make_pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <workload tag>"""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        if "bounce_kernel" not in n:
            continue
        # bounce_kernel<kLds, StackT, kPrimary, kLoop, kMulti, kTex>
        args = n[n.index("bounce_kernel<") + len("bounce_kernel<"):].split(">")[0].split(", ")
        key = "bounce<loop>" if len(args) > 3 and args[3] == "true" else "bounce<wavefront>"
        agg[key][0] += float(r["Counter_Value"])
        agg[key][1] += 1
    return {k: (v / n, n) for k, (v, n) in agg.items()}


def main():
    fetch, write, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    out = {"workload": tag, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 gfx950 correction"}
    for k in f:
        out[k] = {
            "fetch_kb_per_launch_raw": f[k][0], "write_kb_per_launch": w[k][0], "launches": f[k][1],
            "hbm_bytes_per_launch": (2.0 * f[k][0] + w[k][0]) * 1024.0,
        }
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
