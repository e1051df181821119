"""Summarise a rocprofv3 kernel_stats.csv per step and per category. usage: prof_summary.py <dir> <steps_executed>"""
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(fs[0])))
cats = collections.OrderedDict([("conv fwd/dgrad", ("conv_halo", "conv_gather", "conv_ksplit", "conv_ru", "conv_march", "conv_marchg", "conv_lowg", "pointwise_conv", "deconv_fwd")), ("weight grad", ("wgrad_",)),
        ("norm+act", ("normact", "norm_finalize", "channel_stats", "colsum", "amax_", "cast_fp8")), ("layout", ("pack_kernel", "pack_tile_kernel", "unpack_kernel", "wpack")),
        ("pool/loss/adam", ("maxpool", "l1_", "adamw", "gan_gen_loss", "gan_discr_loss")), ("torch native", ("at::native", "rocclr", "Memset", "Cijk")), ])
tot = collections.Counter(); cnt = collections.Counter()
for r in rows:
    name = r["Name"]; c = "other"
    for k, pats in cats.items():
        if any(p in name for p in pats): c = k; break
    tot[c] += int(r["TotalDurationNs"]); cnt[c] += int(r["Calls"])
T = sum(tot.values())
print(f"total {T/n/1e6:.3f} ms/step, {sum(cnt.values())/n:.0f} launches/step")
for c in list(cats) + ["other"]:
    if cnt[c]: print(f"  {c:16s} {tot[c]/n/1e6:7.3f} ms/step {100*tot[c]/T:5.1f}%  {cnt[c]/n:6.1f} launches")
top = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for r in rows[:top]:
    print(f"{int(r['TotalDurationNs'])/n/1e6:8.3f} ms/step {int(r['Calls'])/n:7.1f} calls avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
