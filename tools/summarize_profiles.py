#!/usr/bin/env python3
"""tools/summarize_profiles.py RAW_DIR TAG -- condense the rocprofv3 output of tools/profile_round.sh into the
small text / CSV / JSON files that are committed under profiles/ (written to RAW_DIR/profiles_TAG/).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are
collected in separate passes, are in KB per dispatch, and on gfx950 FETCH_SIZE counts wide (coalesced) reads at half
size -> x2 for the read bytes.  SQ counters are in units of 4 cycles per wavefront (quad-cycles), summed over XCDs."""
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
CLOCK_HZ = 2.4e9     # peak engine clock (MI355X_MICROARCH.md); the busy figures below are fractions of peak issue
NSIMD = 1024


def find(pattern):
    hits = sorted(glob.glob(os.path.join(raw, pattern), recursive=True))
    return hits[0] if hits else None


def trace_stats(wl):
    """Per-kernel statistics from the per-dispatch kernel trace: all launches, and the FULL-SIZE launches only (grid =
    the kernel's largest grid in the run) -- the bench makes a few one-model launches for its synthetic observed data,
    which the plain kernel_stats.csv of rocprofv3 averages in."""
    f = find("trace_%s/**/*kernel_trace.csv" % wl)
    if not f:
        return
    rows = defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if name.startswith("void at::") or "rocclr" in name:
            continue
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        rows[name].append((grid, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"])))
    outp = os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, wl))
    with open(outp, "w") as fo:
        wr = csv.writer(fo, quoting=csv.QUOTE_ALL)
        wr.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs", "FullSizeCalls", "FullSizeAverageNs", "FullSizeMinNs", "FullSizeMaxNs",
                     "FullSizeSteadyAverageNs(after the first 6)", "VGPR_Count", "LDS_Block_Size", "Grid"])
        print("== kernel stats", wl)
        for name, v in sorted(rows.items(), key=lambda kv: -sum(d for _, d, _, _ in kv[1])):
            gmax = max(g for g, _, _, _ in v)
            full = [d for g, d, _, _ in v if g == gmax]
            allv = [d for _, d, _, _ in v]
            steady = full[6:] if len(full) > 8 else full
            wr.writerow([name, len(allv), "%.1f" % (sum(allv) / len(allv)), min(allv), max(allv), len(full), "%.1f" % (sum(full) / len(full)),
                         min(full), max(full), "%.1f" % (sum(steady) / len(steady)), v[0][2], max(l for _, _, _, l in v), gmax])
            print("    %-66s calls %4d  full-size %4d  avg %10.1f us  steady %10.1f us" % (name[:66], len(allv), len(full), sum(full) / len(full) / 1e3,
                  sum(steady) / len(steady) / 1e3))


for wl in ("c2", "c2fast", "c2fastexact", "c3", "c3fast", "c3g", "c2_b65536", "c4", "c5", "rf_c3", "rf_tut", "rf_t512u", "rf_t512r", "rf_n16384", "gauss"):
    trace_stats(wl)
for name in ("latency.txt", "latency_fast.txt", "rf_alone.txt", "gauss_alone.txt", "love_scan.txt", "phase_c3.txt", "phase_c3_fast.txt", "phase_c3_fastexact.txt", "c3_tail.txt", "fuzz_reference.txt", "fuzz_fast.txt",
             "fuzz_lean.txt", "fuzz_lean_prior.txt", "lean_rounds.txt", "lean_guard.txt", "chain_guard.txt", "faeval.txt", "fadiff.txt"):
    if os.path.exists(os.path.join(raw, name)):
        shutil.copy(os.path.join(raw, name), os.path.join(out, "%s_%s" % (tag, name)))
b0 = os.path.join(raw, "bench_default.json")
if os.path.exists(b0) and os.path.getsize(b0):
    shutil.copy(b0, os.path.join(out, "%s_bench_default.json" % tag))
    d0 = json.loads(open(b0).read().strip().splitlines()[-1])
    print("== bench default: c2", round(d0["value"]), "evals/s", round(d0["ms_per_step"], 3), "ms/step;",
          " ".join("%s %s" % (k, round(d0[k]["value"])) for k in ("c3", "c4", "c5") if k in d0 and "value" in d0[k]))
for wl in ("c2p", "c2g", "c3g", "c2_reference", "c2_fastexact", "c2_b65536", "c2_b65536_reference", "c2_b512", "c2_b16384", "c4_arithfast", "c5_arithfast", "c5_full_arithfast", "c4_depth1", "c5_depth1", "c4_reference", "c4_fast_rayleigh", "c4_fast",
           "c5_reference", "c5_fast_rayleigh", "c5_fast"):
    b = os.path.join(raw, "bench_%s.json" % wl)
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(out, "%s_bench_%s.json" % (tag, wl)))
        d = json.loads(open(b).read().strip().splitlines()[-1])
        print("== bench", wl, round(d["value"]), d["unit"], round(d["ms_per_step"], 3), "ms/step")


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


def steady(v):
    """mean over the dispatches of full steps (drops the small set-up launches: < half of the largest)"""
    big = [x for x in v if x >= 0.5 * max(v)]
    return sum(big) / len(big)


summary = {}
DOM = {"c2": ("swd_group_kernel", 4096), "c3": ("swd_group_kernel", 4096), "c2noboard": ("swd_group_kernel", 4096),
       "c2fast": ("swd_lean_kernel", 4096), "c3fast": ("swd_lean_kernel", 4096),   # the engine's defaults: the trial-per-lane kernel
       "c2fastexact": ("swd_group_kernel", 4096),                                   # short refinement, the reference's arithmetic
       "rf_c3": ("rf_synth_kernel", 4096)}


def kernel_ms(wl, dom):
    """the dominant kernel's steady full-size duration in the kernel trace of the same workload [ms]"""
    f = os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, wl))
    if not os.path.exists(f):
        return None
    if wl.startswith("rf_"):   # the receiver function is two kernels per call (coefficients + synthesis): bench.py times both
        tot = sum(float(r["FullSizeSteadyAverageNs(after the first 6)"]) for r in csv.DictReader(open(f)) if "rf_" in r["Name"])
        return tot / 1e6 if tot > 0 else None
    for r in csv.DictReader(open(f)):
        if dom in r["Name"]:
            return float(r["FullSizeSteadyAverageNs(after the first 6)"]) / 1e6
    return None
for wl, (dom, batch) in DOM.items():
    lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of the %s command of tools/profile_round.sh" % wl,
             "units: KB per dispatch (mean over the full-step dispatches); gfx950 correction: FETCH_SIZE x2 for wide coalesced reads "
             "(MI355X_MICROARCH.md, HBM)", ""]
    if "c3" in wl:   # (say what the counter passes of a fused call run)
        lines[2:2] = ["NOTE: under rocprofv3 --pmc the engine runs WITHOUT the start gate between the dispersion kernel and the receiver-function "
                      "stream (csrc/bh_tuning.h: under_pmc -- counter collection serialises the dispatches of all queues): these passes show "
                      "the kernels' own figures, with the regular coefficient kernel (rf_coef_layers_kernel), not their placement beside the "
                      "dispersion kernel and not the `_small` build the kernel trace of the same command shows."]
    traffic = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        for kern, cs in counters("pmc_%s_%s/**/*counter_collection.csv" % (wl, cname)).items():
            if any(k in kern for k in ("swd_", "like_kernel", "rf_", "gauss", "order_", "chain_")):
                v = cs[cname]
                lines.append("%-11s %-64s dispatches %3d  mean %12.3f KB" % (cname, kern[:64], len(v), steady(v)))
                if dom in kern and steady(v) >= traffic.get(cname, 0.0):
                    traffic[cname] = steady(v)
    entry = {"batch": batch, "kernel": dom, "tag": tag, "commit": os.environ.get("BH_PROFILE_COMMIT")}   # (bench.py flags stale entries)
    if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
        hbm = (2.0 * traffic["FETCH_SIZE"] + traffic["WRITE_SIZE"]) * 1024.0
        lines += ["", "%s HBM traffic per launch = 2 x FETCH + WRITE = %.0f bytes" % (dom, hbm)]
        entry.update(fetch_kb=traffic["FETCH_SIZE"], write_kb=traffic["WRITE_SIZE"], hbm_bytes_per_launch=hbm,
                     source="profiles/%s_pmc_hbm_%s.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, FETCH x2 "
                            "gfx950 correction)" % (tag, wl))
    open(os.path.join(out, "%s_pmc_hbm_%s.txt" % (tag, wl)), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    sq = counters("pmc_%s_SQ/**/*counter_collection.csv" % wl)
    for kern, cs in counters("pmc_%s_SQ2/**/*counter_collection.csv" % wl).items():
        for k, v in cs.items():
            sq[kern][k] = v
    lines = ["rocprofv3 --pmc SQ_* (two passes) of the %s command; per full-size dispatch, summed over XCDs; cycle counters in quad-cycles" % wl, ""]
    # (a call may launch several builds of the dominant kernel -- the one-model launches of the bench's synthetic data, the
    #  short refinement's re-run launch: the entry describes the one that issues the most vector instructions)
    heavy = max((k for k in sq if dom in k), key=lambda k: steady(sq[k].get("SQ_INSTS_VALU", [0.0])) * len(sq[k].get("SQ_INSTS_VALU", [0.0])), default=None)
    for kern, cs in sq.items():
        if dom in kern:
            m = {k: steady(v) for k, v in cs.items()}
            lines.append(kern[:100])
            for k in sorted(m):
                lines.append("%-22s %16.0f" % (k, m[k]))
            if kern != heavy:
                lines.append("")
                continue
            kms = kernel_ms(wl, dom)
            if m.get("SQ_WAVE_CYCLES") and m.get("SQ_ACTIVE_INST_VALU"):
                lines.append("VALU-active share of resident-wave cycles = %.3f" % (m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]))
            if m.get("SQ_THREAD_CYCLES_VALU") and m.get("SQ_ACTIVE_INST_VALU"):
                alf = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])
                lines.append("active-lane fraction of VALU cycles (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)) = %.3f" % alf)
                entry["active_lane_frac"] = alf
            if kms and m.get("SQ_ACTIVE_INST_VALU"):
                busy = m["SQ_ACTIVE_INST_VALU"] * 4.0 / (kms * 1e-3 * NSIMD * CLOCK_HZ)
                lines.append("VALU busy over the kernel = SQ_ACTIVE_INST_VALU x 4 / (%.3f ms x %d SIMDs x %.1f GHz) = %.3f" % (kms, NSIMD, CLOCK_HZ / 1e9, busy))
                entry["valu_busy"] = busy
                entry["kernel_ms"] = kms
            if m.get("SQ_INSTS_VALU"):
                entry["valu_insts_per_launch"] = m["SQ_INSTS_VALU"]
                entry["waves_per_launch"] = m.get("SQ_WAVES")
    if len(lines) > 2:
        open(os.path.join(out, "%s_pmc_sq_%s.txt" % (tag, wl)), "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
    summary[wl] = entry
sq = counters("pmc_b65536_SQ/**/*counter_collection.csv")
lines = ["SQ counters of the lane-per-model kernels at B = 65536 (the throughput regime), per dispatch", ""]
for kern, cs in sq.items():
    if "swd_kernel" in kern:
        m = {k: steady(v) for k, v in cs.items()}
        lines.append(kern[:80])
        for k in sorted(m):
            lines.append("   %-22s %16.0f" % (k, m[k]))
        if m.get("SQ_WAVE_CYCLES"):
            lines.append("   VALU-active share of resident-wave cycles = %.3f, active-lane fraction %.3f"
                         % (m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])))
open(os.path.join(out, "%s_pmc_sq_b65536.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
sq = counters("pmc_gauss_SQ/**/*counter_collection.csv")
lines = ["SQ counters of the Gauss-law contraction (tools/gpu_gauss_perf.py 4096 1024), per dispatch", ""]
for kern, cs in sq.items():
    if "gauss_quad" in kern:
        m = {k: steady(v) for k, v in cs.items()}
        lines.append(kern[:90])
        for k in sorted(m):
            lines.append("   %-30s %16.0f" % (k, m[k]))
        kms = kernel_ms("gauss", "gauss_quad")
        if kms:
            lines.append("   kernel %.4f ms (kernel trace, steady) -> 2 B n^2 / t = %.1f TFLOP/s = %.1f %% of the 78.6 TFLOP/s FP64 matrix peak"
                         % (kms, 2.0 * 4096 * 1024 * 1024 / (kms * 1e-3) / 1e12, 2.0 * 4096 * 1024 * 1024 / (kms * 1e-3) / 1e12 / 78.6 * 100))
open(os.path.join(out, "%s_pmc_sq_gauss.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
