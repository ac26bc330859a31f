#!/usr/bin/env python3
"""Summaries of rocprofv3 output keyed by kernel AND launch shape (blocks = Grid_Size / Workgroup_Size).

  pmc_summary.py pmc <out.json> name=dir [name=dir ...]     counter passes (e.g. fetch=gpurun_out/r05/fetch write=...)
  pmc_summary.py trace <out.json> <dir>                      kernel-trace pass: average duration per (kernel, blocks)

A kernel that is launched with several grid sizes in one profiled command (the 239-block headline job and the 1 024-block
throughput job of bench.py) gets one entry per grid size: "<kernel>@<blocks>", beside the all-launch entry "<kernel>"
(an average over two launch shapes describes no launch that exists)."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    return name.split("(")[0]


def pmc(out, pairs):
    res = {}
    for pair in pairs:
        key, d = pair.split("=", 1)
        agg = collections.defaultdict(lambda: [0.0, 0])
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(fn)):
                k = short(r["Kernel_Name"])
                if "tsq" not in k:
                    continue
                blocks = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
                v = float(r["Counter_Value"])
                for kk in (k, f"{k}@{blocks}"):
                    agg[(kk, r["Counter_Name"])][0] += v
                    agg[(kk, r["Counter_Name"])][1] += 1
        names = sorted({c for (_, c) in agg})
        table = {}
        for (k, c), (s, n) in sorted(agg.items()):
            e = {"sum": s, "dispatches": n, "per_dispatch": s / max(n, 1)}
            if len(names) == 1:
                table[k] = e
            else:
                table.setdefault(k, {})[c] = e
        res[key] = table
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import turbosqueeze_amd
    res["kernel_fingerprint"] = turbosqueeze_amd.source_fingerprint()
    res["keys"] = "\"<kernel>\" = every launch of the profiled command; \"<kernel>@<blocks>\" = the launches with that many workgroups"
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def trace(out, d):
    agg = collections.defaultdict(list)
    for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            k = short(r["Kernel_Name"])
            if "tsq" not in k:
                continue
            blocks = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
            agg[f"{k}@{blocks}"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    res = {k: {"launches": len(v), "avg_ms": sum(v) / len(v), "min_ms": min(v), "max_ms": max(v)} for k, v in sorted(agg.items())}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3:])
    else:
        trace(sys.argv[2], sys.argv[3])
