"""Timeline of the input feed under hipGraph replay: memory copies vs kernel activity (rocprofv3 --kernel-trace --memory-copy-trace, csv).
usage: feed_trace.py <dir>"""
import csv, glob, sys
d = sys.argv[1]
k = sorted(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
m = sorted(csv.DictReader(open(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
t_end = int(k[-1]["End_Timestamp"])
win0 = t_end - 60_000_000                      # the last 60 ms: ~4 steady steps
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in k if int(r["Start_Timestamp"]) >= win0]
ms = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind", "?")), r.get("Size", r.get("Bytes", "?"))) for r in m if int(r["End_Timestamp"]) >= win0]
print(f"{len(ks)} kernels, {len(ms)} copies in the last 60 ms")
for s, e, dr, sz in ms:
    if e - s > 20_000:
        print(f"copy {dr:28s} {sz:>12s} B  start {(s - win0) / 1e6:8.3f} ms  dur {(e - s) / 1e6:7.3f} ms")
# kernel-idle gaps > 30 us
prev = ks[0][1]
busy = 0
for s, e, n in ks:
    if s - prev > 30_000:
        print(f"gap {(s - prev) / 1e3:8.1f} us at {(prev - win0) / 1e6:8.3f} ms before {n[:60]}")
    busy += e - max(s, prev) if e > prev else 0
    prev = max(prev, e)
print(f"kernel-busy {busy / 1e6:.3f} ms of {(ks[-1][1] - ks[0][0]) / 1e6:.3f} ms")
# adamw launches mark the phases: print their start times
for s, e, n in ks:
    if "adamw" in n:
        print(f"adamw at {(s - win0) / 1e6:8.3f} ms")
