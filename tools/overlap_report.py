"""Do RCCL's kernels run CONCURRENTLY with the backward kernels of the segment after them?  usage: overlap_report.py <rocprofv3 dir>
Reads the kernel trace of `bench.py --force-collectives` (one rank, a real RCCL group, the five-segment step with its four all-reduces
between the graph replays) and prints, for every collective kernel of the last step, which compute kernels overlap it in time."""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
is_coll = lambda n: any(k in n for k in ("nccl", "rccl", "AllReduce", "ncclDevKernel"))
colls = [e for e in ev if is_coll(e[2])]
print(f"# {len(ev)} dispatches, {len(colls)} collective kernels")
if not colls:
    print("no collective kernel in the trace"); sys.exit(0)
last = colls[-4:] if len(colls) >= 4 else colls            # the last step's four all-reduces
for s, e, name in last:
    over = [(max(s, a) , min(e, b), n) for a, b, n in ev if not is_coll(n) and a < e and b > s]
    tot = sum(y - x for x, y, _ in over)
    print(f"collective {name[:60]:60s} {(e - s) / 1e3:9.1f} us; compute kernels overlapping it: {len(over)}, overlapped time {tot / 1e3:9.1f} us")
    for x, y, n in over[:6]:
        print(f"      {(y - x) / 1e3:8.1f} us  {n[:90]}")
    # the kernels right before and after it on the timeline
    before = [n for a, b, n in ev if b <= s and not is_coll(n)][-1:]
    after = [n for a, b, n in ev if a >= e and not is_coll(n)][:1]
    print(f"      previous compute kernel: {before[0][:70] if before else '-'}; next: {after[0][:70] if after else '-'}")
