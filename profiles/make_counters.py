"""Summarises the rocprofv3 PMC passes of a profiles/collect*.sh run into <dir>/counters.json: per kernel class, per launch --
HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, in KB: MI355X_MICROARCH.md's gfx950 correction for FETCH_SIZE), wave64 VALU instructions,
waves, and where a wave's cycles go (SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES).  Stamped with the
workload tag and the hash of the kernel sources (tools/source_hash.py); bench.py merges it into its roofline object only while both
match the run.

usage: python profiles/make_counters.py <dir> <workload tag> <frames per pass>     (dir holds fetch/ write/ sq/ [tcp/ tcc/] as collect.sh makes them)"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from source_hash import kernel_knobs, kernel_source_hash  # noqa: E402


def kernel_class(name):
    if "bounce_kernel<" in name:
        a = name[name.index("bounce_kernel<") + len("bounce_kernel<"):].split(">")[0].split(", ")  # kLds, StackT, kPrimary, kLoop, ...
        return "bounce<loop>" if a[3] == "true" else ("bounce<primary>" if a[2] == "true" else "bounce<compact>")
    for key, cls in (("primary_kernel<", "primary"), ("traverse_dyn_kernel<", "traverse"), ("traverse_kernel<", "traverse"), ("tail_kernel<", "tail"),
                     ("shade_kernel<", "shade"), ("di_kernel<", "di"), ("beam_kernel<", "beams")):
        if key in name:
            return cls
    return None


def main():
    d, tag, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kernel_class(r["Kernel_Name"])
            if k is None:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    out = {"workload": tag, "kernel_source_hash": kernel_source_hash(), "knobs": kernel_knobs(), "frames_in_flight_of_the_counter_passes": 1, "frames_per_pass": frames,
           "source": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 bench.py ... --frames-in-flight 1 (profiles/collect*.sh); "
                     "FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md)", "kernels": {}}
    for k, c in agg.items():
        per = lambda name: c[name] / cnt[k][name] if cnt[k][name] else None
        e = {"launches": int(min(cnt[k].values()))}  # (every counter but SQ_WAVES is collected in exactly one pass)
        if per("FETCH_SIZE") is not None and per("WRITE_SIZE") is not None:
            e.update(fetch_kb_per_launch_raw=per("FETCH_SIZE"), write_kb_per_launch=per("WRITE_SIZE"), hbm_bytes_per_launch=(2.0 * per("FETCH_SIZE") + per("WRITE_SIZE")) * 1024.0)
        if per("SQ_WAVES"):
            wc = per("SQ_WAVE_CYCLES")
            e.update(waves_per_launch=per("SQ_WAVES"), valu_insts_per_launch=per("SQ_INSTS_VALU"), valu_insts_per_wave=per("SQ_INSTS_VALU") / per("SQ_WAVES"),
                     active_frac=per("SQ_ACTIVE_INST_ANY") / wc, wait_any_frac=per("SQ_WAIT_ANY") / wc, wait_inst_frac=per("SQ_WAIT_INST_ANY") / wc)
            if per("SQ_INSTS_VMEM") is not None:
                e["vmem_insts_per_launch"] = per("SQ_INSTS_VMEM")
            if per("SQ_ACTIVE_INST_VALU") is not None:
                e["valu_active_frac_of_wave"] = per("SQ_ACTIVE_INST_VALU") / wc
        for name in ("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum"):
            if per(name) is not None:
                e[name.lower() + "_per_launch"] = per(name)
        out["kernels"][k] = e
    # frames of a pass = launches of the kernel every frame starts with (bench.py renders more frames than its --steps: warm-up, the
    # event-bracketed region, the one-frame-at-a-time frames), so the count comes from the trace, not from the command line
    for first in ("bounce<primary>", "primary"):
        if first in out["kernels"]:
            out["frames_per_pass"] = out["kernels"][first]["launches"]
            break
    json.dump(out, open(os.path.join(d, "counters.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
