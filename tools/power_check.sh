#!/bin/bash
# Operand bit activity vs kernel time AND clock: tests/diag/diag_power.py under rocprofv3 --pmc (GRBM_GUI_ACTIVE, SQ counters).
# usage: tools/power_check.sh <tag>     (on the GPU box) -> gpurun_out/<tag>_power_check.txt
tag=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${tag}_power
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT -o p -- python3 $ROOT/tests/diag/diag_power.py > $OUT/run.log 2>&1
echo "rc=$?"
cd $ROOT
python3 - <<PY > gpurun_out/${tag}_power_check.txt
import csv, glob, collections
out = "$OUT"
tr = {r["Dispatch_Id"]: r for r in csv.DictReader(open(glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0])) if "conv_march_kernel" in r["Kernel_Name"]}
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0])):
    if r["Dispatch_Id"] in tr:
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(tr, key=lambda i: int(tr[i]["Start_Timestamp"]))
n = len(ids) // 4
print("# tests/diag/diag_power.py under rocprofv3 --pmc (tools/power_check.sh): conv_march_kernel<false>, 32->32 at 128^3, bf16; the SAME")
print("# instruction stream on operands of different bit activity; per set the mean over its last 100 launches.")
print("# clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)")
print("# (profiled passes run a few per cent slower than un-profiled ones: MI355X_MICROARCH.md, DVFS give-back item 2)")
for g, name in enumerate(("zeros", "const 1.0", "randn", "zeros again")):
    sel = ids[g * n + 3: (g + 1) * n]
    dur = sum(int(tr[i]["End_Timestamp"]) - int(tr[i]["Start_Timestamp"]) for i in sel) / len(sel) / 1e3
    gui = sum(cnt[i].get("GRBM_GUI_ACTIVE", 0) for i in sel) / len(sel) / 8
    mf = sum(cnt[i].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for i in sel) / len(sel) / 1024
    wv = sum(cnt[i].get("SQ_WAVE_CYCLES", 0) for i in sel) / len(sel) * 4 / 1024
    wa = sum(cnt[i].get("SQ_WAIT_ANY", 0) for i in sel) / len(sel) * 4 / 1024
    print(f"{name:12s} {dur:8.1f} us   GUI-active {gui / 1e3:7.1f} k cycles -> {gui / dur / 1e3:5.2f} GHz   MFMA busy {mf / 1e3:6.1f} k cycles = {100 * mf / gui:4.1f} % of the kernel"
          f"   wave lifetime {wv / 1e3:6.1f} k cycles, of which waiting on counters / barriers {100 * wa / max(wv, 1):4.1f} %")
PY
cat gpurun_out/${tag}_power_check.txt
