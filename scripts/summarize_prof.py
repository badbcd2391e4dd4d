"""Condense a scripts/profile_bench.sh output directory into profiles/<name>.* files:
the rocprofv3 --kernel-trace --stats CSV as it is and a text table of the PMC counters
(mean per dispatch of the SA kernel) with the derived per-launch figures.
python scripts/summarize_prof.py gpurun_out/prof_TAG NAME [entries per launch]"""
import collections, csv, glob, os, shutil, sys

src, name = sys.argv[1], sys.argv[2]
# db entries per launch: a workgroup may hold several entries (sat_last_launch_info), so the dispatch's
# grid alone does not say; default = the bench shard
entries_per_launch = int(sys.argv[3]) if len(sys.argv) > 3 else 125000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, name + "_kernel_stats.csv"))
lines = []
kern_ns = None
kern_name = None
if stats:
    for r in csv.DictReader(open(stats[0])):
        if "sat_sa_" in r["Name"]:
            kern_ns = float(r["AverageNs"])
            kern_name = r["Name"].split("(")[0].replace("void ", "")
            lines.append(f"kernel {r['Name']}: calls {r['Calls']} avg {kern_ns/1e6:.3f} ms min {float(r['MinNs'])/1e6:.3f} max {float(r['MaxNs'])/1e6:.3f}")
vals = {}
for p in sorted(glob.glob(os.path.join(src, "pmc*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(p)):
        if "sat_sa_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
    for k, v in agg.items():
        vals[k] = sum(v) / len(v)
    if meta and "meta" not in vals:
        vals["meta"] = {k: meta[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size") if k in meta}
lines.append("PMC counters, mean per dispatch of the SA kernel (separate --pmc passes):")
for k, v in vals.items():
    if k != "meta":
        lines.append(f"  {k:24s} {v:.6g}")
if "meta" in vals:
    lines.append(f"  dispatch: {vals['meta']}")
if "SQ_WAVES" in vals and "SQ_INSTS_VALU" in vals:
    w = vals["SQ_WAVES"]
    lines.append(f"derived: VALU instr / wave / SA step = {vals['SQ_INSTS_VALU']/w/100:.1f}, LDS instr / wave / step = {vals['SQ_INSTS_LDS']/w/100:.1f}, SALU = {vals['SQ_INSTS_SALU']/w/100:.1f}")
if "GRBM_GUI_ACTIVE" in vals and kern_ns:
    cyc = vals["GRBM_GUI_ACTIVE"] / 8
    lines.append(f"derived: shader clock ~ {cyc/kern_ns:.2f} GHz ({cyc:.4g} cycles per launch)")
    if "SQ_INSTS_VALU" in vals:
        lines.append(f"derived: VALU wave-instr per SIMD-cycle = {vals['SQ_INSTS_VALU']/1024/cyc:.3f} (peak 0.5: one wave64 instr per 2 cycles on a SIMD-32)")
    if "SQ_LDS_IDX_ACTIVE" in vals:
        lines.append(f"derived: LDS busy = {vals['SQ_LDS_IDX_ACTIVE']/256/cyc:.3f} of CU cycles, of which bank conflicts {vals['SQ_LDS_BANK_CONFLICT']/vals['SQ_LDS_IDX_ACTIVE']:.3f}")
    if "SQ_WAVE_CYCLES" in vals:
        lines.append(f"derived: resident waves ~ {vals['SQ_WAVE_CYCLES']*4/cyc/1024:.2f} per SIMD")
if "FETCH_SIZE" in vals:
    fetch = vals["FETCH_SIZE"] * 1024 * 2     # KiB units; gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md HBM)
    wr = vals.get("WRITE_SIZE", 0) * 1024
    lines.append(f"derived: HBM traffic per launch = {fetch/1e6:.1f} MB read (FETCH_SIZE x 1024 x 2) + {wr/1e6:.2f} MB written")
if "FETCH_SIZE" in vals:
    import json
    sys.path.insert(0, root)
    from cuda_satabsearch_amd import build
    json.dump({"source": name + "_summary.txt", "entries_per_launch": entries_per_launch,
               # which kernel and which sources the figures belong to: bench.py drops them when either differs
               "kernel": kern_name, "kernel_source_sha256": build.kernel_source_hash(),
               "fetch_size_kib": vals["FETCH_SIZE"], "write_size_kib": vals.get("WRITE_SIZE", 0.0),
               "hbm_bytes_per_launch": vals["FETCH_SIZE"] * 1024 * 2 + vals.get("WRITE_SIZE", 0.0) * 1024,
               "valu_wave_instr_per_launch": vals.get("SQ_INSTS_VALU"), "lds_wave_instr_per_launch": vals.get("SQ_INSTS_LDS"),
               "kernel_ms_profiled": (kern_ns / 1e6) if kern_ns else None,
               "lds_busy_frac": (vals["SQ_LDS_IDX_ACTIVE"] / 256 / (vals["GRBM_GUI_ACTIVE"] / 8)) if "SQ_LDS_IDX_ACTIVE" in vals and "GRBM_GUI_ACTIVE" in vals else None,
               "note": "FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE x 1024; separate --pmc passes"},
              open(os.path.join(dst, "bench_traffic.json"), "w"), indent=1)
open(os.path.join(dst, name + "_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
