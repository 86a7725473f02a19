"""Per-launch timeline of the last frame in a rocprofv3 kernel trace: python tools/ktimeline.py <dir> <frames>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
frames = int(sys.argv[2])
rows = [r for r in csv.DictReader(open(f)) if "k_wf" in r["Kernel_Name"] or "k_dof" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // frames
t0 = int(rows[(frames - 1) * n]["Start_Timestamp"])
for r in rows[(frames - 1) * n:]:
    print(f'{r["Kernel_Name"].split("(")[0][:24]:24s} start {(int(r["Start_Timestamp"])-t0)/1e6:8.3f} dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6:7.3f} ms  blocks {int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"])}')
