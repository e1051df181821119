"""Busy vs idle time of the GPU from a rocprofv3 kernel trace: python tools/trace_gaps.py <dir> [last_n_kernels]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
span = rows[-1][1] - rows[0][0]
busy = 0; cur_end = rows[0][0]; gaps = []
for s, e, k in rows:
    if s > cur_end:
        gaps.append((s - cur_end, k))
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
print(f"kernels {len(rows)} span {span/1e6:.3f} ms busy {busy/1e6:.3f} ms idle {100*(span-busy)/span:.1f}%  mean gap {sum(g for g,_ in gaps)/max(1,len(gaps))/1e3:.2f} us over {len(gaps)} gaps")
gaps.sort(reverse=True)
for g, k in gaps[:12]:
    print(f"  gap {g/1e3:8.1f} us before {k[:80]}")
