"""Copies what tools/profile_round.sh left in gpurun_out/prof_<tag>/ into profiles/<tag>_* (tracked):
the bench line, the rocprofv3 kernel statistics, the PMC passes cut down to this library's kernels
(k_*), their digest, and refreshes profiles/traffic.json.   python tools/save_profiles.py <tag>"""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
shutil.copy(os.path.join(src, "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
shutil.copy(os.path.join(src, "digest.json"), os.path.join(dst, tag + "_pmc_digest.json"))
for f in sorted(os.listdir(src)):
    if f.startswith("pmc_") and f.endswith(".csv"):
        rows = list(csv.reader(open(os.path.join(src, f))))
        name = rows[0].index("Kernel_Name")
        keep = [rows[0]] + [r for r in rows[1:] if r[name].startswith("k_") or " k_" in r[name]]
        csv.writer(open(os.path.join(dst, tag + "_" + f, ), "w", newline="")).writerows(keep)
dig = json.load(open(os.path.join(src, "digest.json")))
tr_path = os.path.join(dst, "traffic.json")
tr = json.load(open(tr_path))
h = dig.get("fabric_bytes_per_launch") or dig["hbm_bytes_per_launch"]
hist = tr.setdefault("history", {})
hist[tag] = {"read_raw": h["read_raw"], "write": h["write"], "total_corrected": h["total_corrected"]}
tr["kernel"] = f"k_wf_pipeline<false, false>, final.obj 3840x2160 MSAA4, one launch = one frame (profiles/{tag}_pmc_fetch.csv, {tag}_pmc_write.csv)"
tr["FETCH_SIZE_KB_per_launch_raw"] = dig["counters"]["FETCH_SIZE"]
tr["WRITE_SIZE_KB_per_launch_raw"] = dig["counters"]["WRITE_SIZE"]
tr["k_wf_pipeline_hbm_bytes_per_launch"] = int(h["total_corrected"])
tr["k_wf_pipeline_fabric_bytes_per_launch"] = int(h["total_corrected"])
if dig.get("l2_to_memory_requests"):
    tr["l2_to_memory_requests_per_launch"] = dig["l2_to_memory_requests"]
tr["k_wf_pipeline_hbm_bytes_per_launch_uncorrected"] = int(h["read_raw"] + h["write"])
json.dump(tr, open(tr_path, "w"), indent=1)
print("saved", tag, int(h["total_corrected"]))
