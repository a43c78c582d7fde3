#!/usr/bin/env python3
"""tools/summarize_profiles.py RAW_DIR TAG -- condense the rocprofv3 output of tools/profile_round.sh
into the small text/CSV/JSON files that are committed under profiles/ (written to RAW_DIR/profiles_TAG/).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are collected in separate passes, are in KB per dispatch, and on gfx950 FETCH_SIZE counts
wide (coalesced) reads at half size -> x2 for the read bytes."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

raw, tag = sys.argv[1], sys.argv[2]
out = os.path.join(raw, "profiles_" + tag)
os.makedirs(out, exist_ok=True)


def find(pattern):
    hits = sorted(glob.glob(os.path.join(raw, pattern), recursive=True))
    return hits[0] if hits else None


for wl in ("c2", "c3", "c2g", "c3g"):
    f = find("trace_%s/**/*kernel_stats.csv" % wl)
    if f:
        shutil.copy(f, os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, wl)))
        print("== kernel stats", wl)
        print(open(f).read())
    b = os.path.join(raw, "bench_%s.json" % wl)
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(out, "%s_bench_%s.json" % (tag, wl)))
        d = json.loads(open(b).read().strip().splitlines()[-1])
        print("== bench", wl, d["value"], "evals/s", d["ms_per_step"], "ms/step", d["kernel_ms_per_step"])


def counters(pattern):
    """-> {kernel: {counter: [values per dispatch]}}"""
    f = find(pattern)
    res = defaultdict(lambda: defaultdict(list))
    if not f:
        return res
    per_dispatch = defaultdict(float)
    names = {}
    for row in csv.DictReader(open(f)):
        key = (row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[key] += float(row["Counter_Value"])       # summed over XCDs / instances
        names[row["Dispatch_Id"]] = row["Kernel_Name"]
    for (did, cname), v in per_dispatch.items():
        res[names[did]][cname].append(v)
    return res


lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --no-cpu-baseline --steps 4 --warmup 1"
         " (c2, B=4096, L=10, K=30)",
         "units: KB per dispatch; gfx950 correction: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM)", ""]
traffic = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    for kern, cs in counters("pmc_%s/**/*counter_collection.csv" % cname).items():
        if "swd_" in kern or "like_kernel" in kern or "rf_" in kern or "gauss" in kern:
            v = cs[cname]
            mean = sum(v) / len(v)
            lines.append("%-11s %-70s dispatches %3d  mean %12.3f KB" % (cname, kern[:70], len(v), mean))
            if "swd_group_kernel" in kern:
                traffic[cname] = mean
if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
    hbm = (2.0 * traffic["FETCH_SIZE"] + traffic["WRITE_SIZE"]) * 1024.0
    lines.append("")
    lines.append("swd_group_kernel HBM traffic per launch = 2 x FETCH + WRITE = %.0f bytes" % hbm)
    json.dump({"workload": "c2", "batch": 4096, "kernel": "swd_group_kernel", "fetch_kb": traffic["FETCH_SIZE"],
               "write_kb": traffic["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
               "source": "profiles/%s_pmc_hbm.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, "
                         "FETCH x2 gfx950 correction)" % tag},
              open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
open(os.path.join(out, "%s_pmc_hbm.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

sq = counters("pmc_SQ/**/*counter_collection.csv")
lines = ["rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -- python bench.py "
         "--no-cpu-baseline --steps 4 --warmup 1 (c2); per dispatch, summed over XCDs", ""]
for kern, cs in sq.items():
    if "swd_group_kernel" in kern:
        m = {k: sum(v) / len(v) for k, v in cs.items()}
        for k in sorted(m):
            lines.append("%-22s %16.0f" % (k, m[k]))
        if m.get("SQ_WAVE_CYCLES") and m.get("SQ_ACTIVE_INST_VALU"):
            lines.append("VALU-active share of resident-wave cycles = %.3f" % (m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]))
        if m.get("SQ_THREAD_CYCLES_VALU") and m.get("SQ_ACTIVE_INST_VALU"):
            lines.append("active-lane fraction of VALU cycles (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)) = %.3f"
                         % (m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])))
open(os.path.join(out, "%s_pmc_sq.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
