"""Per-kernel HBM-side bytes per launch from two rocprofv3 --pmc passes (tools/pmc_traffic.sh).
Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE
tallies the 128-byte requests of 16-byte-per-lane streams (global_load_dwordx4, buffer_load ... lds -- every operand
stream of these kernels) at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections, csv, glob, json, os, subprocess, sys

out, tag, size, dtype = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]][name].append(float(r["Counter_Value"]))
rows = []
for k, cs in acc.items():
    f, w = cs.get("FETCH_SIZE", []), cs.get("WRITE_SIZE", [])
    n = max(len(f), len(w), 1)
    fb, wb = 2.0 * 1024.0 * sum(f) / max(1, len(f)), 1024.0 * sum(w) / max(1, len(w))
    rows.append((n * (fb + wb), k, n, fb, wb))
rows.sort(reverse=True)
lines = [f"# HBM-side bytes per launch, eager GAN step at 1x24x{size}^3, {dtype} (tools/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate",
         "# rocprofv3 --pmc passes; read = 2 x FETCH_SIZE (gfx950: 16-B/lane streams are tallied at half their bytes), write = WRITE_SIZE)",
         "# kernel | launches in the profiled steps | read MB/launch | write MB/launch | share of all bytes"]
tot = sum(r[0] for r in rows) or 1.0
for t, k, n, fb, wb in rows[:30]:
    lines.append(f"{k[:100]:100s} {n:5d} {fb / 1e6:10.2f} {wb / 1e6:10.2f} {100 * t / tot:6.1f}%")
for dst in (os.path.join(root, "profiles"), out):       # (the GPU box returns gpurun_out/ only: copy from there into profiles/)
    open(os.path.join(dst, f"{tag}_{dtype}_{size}_pmc_step_traffic.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:18]))

import bench  # noqa: E402  (kernel_source_hash, PLAN_NAMES)
entries = []
for t, k, n, fb, wb in rows:
    for plain in set(bench.PLAN_NAMES.values()):
        base = plain.split("<")[0]
        if base in k and (("<" not in plain) or plain.replace(" ", "") in k.replace(" ", "")):
            nm = plain + ("<e4m3>" if "<true>" in k and base == "conv_march_kernel" else "")
            if base == "conv_march_kernel" and dtype == "fp8" and "<true>" not in k:
                continue
            entries.append(dict(kernel=nm, dtype=dtype, size=size, workload="gan_step", batch=1, launches=n,
                                hbm_bytes_per_launch=fb + wb, read_bytes_per_launch=fb, write_bytes_per_launch=wb, rocprof_name=k[:120]))
try:
    head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=root).decode().strip()
except Exception:
    head = None
rec = dict(kernel_source_sha16=bench.kernel_source_hash(), git_head=head,
           method="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --no-graph --steps 2; 2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes",
           entries=entries)
for dst in (os.path.join(root, "profiles"), out):
    path = os.path.join(dst, f"{tag}_{dtype}_{size}_traffic.json")
    json.dump(rec, open(path, "w"), indent=1)
    print("wrote", path, len(entries), "entries")
