"""rocprofv3 (ROCm 7: rocpd sqlite output) -> the kernel_stats.csv layout tools/prof_summary.py reads.
usage: prof_db_to_csv.py <results.db> <out_kernel_stats.csv>"""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"))
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, total, avg, mn, mx in rows:
        w.writerow([name, calls, int(total), f"{avg:.1f}", f"{100.0 * total / tot:.3f}", int(mn), int(mx)])
print(f"{len(rows)} kernels, {sum(r[1] for r in rows)} dispatches, {tot / 1e6:.3f} ms")
