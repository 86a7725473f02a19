"""Summarise a rocprofv3 kernel trace CSV: python tools/kstats.py <dir> [frames]"""
import csv, glob, sys, collections
d = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:34]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += dur; a[2] = max(a[2], dur)
tot = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:36s} calls/frame {a[0]/frames:7.1f}  ms/frame {a[1]/frames:8.3f}  max {a[2]:7.3f}  {100*a[1]/tot:5.1f}%")
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
print(f"sum {tot/frames:.3f} ms/frame; span {(t1-t0)/1e6/frames:.3f} ms/frame")
