"""Digest of the PMC passes tools/profile_round.sh leaves in a directory: per launch of the
timed pipeline kernel (k_wf_pipeline<false, false>), counters averaged over its launches, plus
the derived figures the design notes quote (lanes active, VALU busy, the bytes between L2 and the memory
side of the fabric with the gfx950 correction of the FETCH_SIZE counter, cache hit rates)."""
import csv, glob, json, os, sys
d = sys.argv[1]
vals = {}
for f in sorted(glob.glob(os.path.join(d, "pmc_*.csv"))):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_wf_pipeline<false" not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        vals[k] = sum(v) / len(v)
out = {"kernel": "k_wf_pipeline<false, false>, final.obj 3840x2160 MSAA4, one launch = one frame", "counters": vals}
g = vals.get
if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    out["lanes_active_in_valu"] = round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")), 4)
if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
    # a wave instruction occupies its SIMD for 4 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs, 1024 SIMDs in all
    out["valu_busy"] = round(g("SQ_ACTIVE_INST_VALU") * 4.0 / (g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0), 4)
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    rd, wr = g("FETCH_SIZE") * 1024.0, g("WRITE_SIZE") * 1024.0
    out["fabric_bytes_per_launch"] = {"read_raw": rd, "read_corrected_x2": 2 * rd, "write": wr, "total_corrected": 2 * rd + wr,
                                   "correction": "gfx950: FETCH_SIZE tallies 128-B requests as 64 B; read side doubled, WRITE_SIZE exact",
                                   "what": "bytes between the L2s and the memory side of the fabric (Infinity Cache + HBM); an upper bound of the HBM bytes"}
if g("TCC_EA0_RDREQ_sum") and g("TCC_EA0_WRREQ_sum"):
    # Where the L2's misses go.  *_DRAM counts the requests addressed to the local memory controllers (as opposed to
    # another device or the host); the Infinity Cache sits in front of those controllers, on the memory side of the
    # fabric, so a request it serves is counted here like one that reaches HBM: these counters bound the HBM bytes
    # from above and cannot say how far above.
    out["l2_to_memory_requests"] = {
        "read": g("TCC_EA0_RDREQ_sum"), "read_for_dram": g("TCC_EA0_RDREQ_DRAM_sum"),
        "write": g("TCC_EA0_WRREQ_sum"), "write_for_dram": g("TCC_EA0_WRREQ_DRAM_sum"),
        "what": "every read and write request the L2 sends on is addressed to local memory (none to a peer or the host); "
                "Infinity-Cache hits are among them - no counter of this stack separates them from HBM accesses"}
if g("TCC_HIT_sum") and g("TCC_REQ_sum"):
    out["l2_hit_rate"] = round(g("TCC_HIT_sum") / g("TCC_REQ_sum"), 4)
if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum"):
    out["l1_hit_rate_reads"] = round(1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum"), 4)
    out["l1_miss_latency_cycles"] = round(g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum"), 1)
print(json.dumps(out, indent=1))
