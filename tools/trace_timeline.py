#!/usr/bin/env python3
"""Timeline of the kernels of ONE steady step from a rocprofv3 --kernel-trace CSV: start / end relative to the dispersion kernel's
start, duration, grid -- what shows whether the receiver-function kernels run beside the dispersion kernel or after it.
    python tools/trace_timeline.py <dir with *_kernel_trace.csv> [kernel name that anchors a step = swd_lean]"""
import csv, glob, os, sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "swd_lean"
f = (glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True) or [None])[0]
if f is None:
    sys.exit("no kernel trace under " + d)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X") or r.get("Grid_Size") or "") for r in csv.DictReader(open(f)))
big = max(int(e[3] or 0) for e in ev if anchor in e[2])
idx = [i for i, e in enumerate(ev) if anchor in e[2] and int(e[3] or 0) == big]
if len(idx) < 4:
    sys.exit("fewer than four full-size launches of " + anchor)
i0, i1 = idx[-3], idx[-2]
t0 = ev[i0][0]
print("kernels of one steady step (the third-last full-size %s launch to the next), microseconds from that launch's start" % anchor)
print("%10s %10s %10s  %s" % ("start", "end", "duration", "kernel (grid)"))
for e in ev[max(0, i0 - 3):i1 + 1]:
    name = e[2].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%10.1f %10.1f %10.1f  %s (%s)" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, name[:70], e[3]))
