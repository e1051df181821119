"""Per-launch listing of ONE eager training step from a rocprofv3 kernel trace (csv).
usage: step_trace.py <dir with *_kernel_trace.csv> [out.txt]
The trace holds several identical steps back to back: the period K is the smallest K with names[-K:] == names[-2K:-K]."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
K = None
for k in range(50, len(names) // 2):
    if names[-k:] == names[-2 * k:-k]:
        K = k
        break
if K is None:
    sys.exit("no repeating step found")
step = rows[-K:]
t0 = int(step[0]["Start_Timestamp"])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
tot = 0
print(f"# {K} launches per step; span {(int(step[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms (eager: includes launch gaps)", file=out)
for i, r in enumerate(step):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    short = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    short = short.split("(")[0][:70]
    g = f"{r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')}x{r.get('Grid_Size_Z', '?')}"
    print(f"{i:4d} {(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  {d:8.1f} us  grid {g:>16s}  {short}", file=out)
print(f"# sum of kernel durations {tot / 1e3:.3f} ms", file=out)
