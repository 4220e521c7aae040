#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag> directory (made by profiles/collect.sh) into the small files that are
committed under profiles/:  <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, our kernels and
the largest torch kernels), <tag>_step_timeline.txt (one steady-state step), <tag>_pmc_traffic.csv and
pmc_traffic.json (HBM bytes per launch of each kernel from FETCH_SIZE / WRITE_SIZE, corrected as
MI355X_MICROARCH.md 'HBM' prescribes: both counters are in KiB... FETCH_SIZE under-reports wide coalesced
reads by 2x on gfx950, WRITE_SIZE is exact for 16-B stores and float atomics).
usage: python profiles/summarize.py <raw dir of collect.sh> <tag> [output dir, default profiles/]"""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
script_dir = os.path.dirname(os.path.abspath(__file__))
here = sys.argv[3] if len(sys.argv) > 3 else script_dir     # where the summary files go (default: profiles/)
sys.path.insert(0, script_dir)
from source_id import source_id  # noqa: E402


def run_identity():
    """What the counters belong to: the kernel sources (source_id.py) and the workload of the traced bench run (P, the
    instance count R of its last view).  bench.py only combines these counters with a live duration when both match."""
    ident = {"tag": tag, "source_id": source_id(os.path.dirname(script_dir))}
    try:
        line = [x for x in open(os.path.join(src, "bench_trace.json")).read().splitlines() if x.startswith("{")][-1]
        j = json.loads(line)
        ident["P"] = j["config"]["gaussians"]
        ident["R"] = j["config"]["num_rendered_last_view"]
        ident["workload"] = j["config"]["workload"]
    except Exception as e:   # noqa: BLE001
        ident["identity_error"] = repr(e)
    return ident
OURS = ("preprocess_fwd_kernel", "scan_block_sums_kernel", "rs_hist_kernel", "rs_scatter_kernel", "scan_reduce_kernel",
        "scan_sums_kernel", "scan_down_kernel", "rs_rowscan_kernel", "sorted_block_sums_kernel", "duplicate_kernel", "tile_ranges_kernel",
        "bin_prepare_kernel", "render_fwd_wave_kernel", "render_bwd_wave_kernel", "render_fwd_kernel", "render_bwd_kernel", "adam_kernel", "scan_small_kernel", "preprocess_bwd_kernel", "l1_fwd_kernel",
        "l1_bwd_kernel", "dwt2_l1_fwd_kernel", "dwt2_l1_bwd_kernel", "ssim_fwd_kernel", "ssim_bwd_kernel",
        "patch_dwt_kernel", "lgdwt_combine_kernel", "act_fwd_kernel", "act_bwd_kernel", "densify_stats_kernel", "patch_means_kernel", "elf_low_kernel", "bilinear_up_kernel", "knn_search_kernel", "preprocess_bwd_step_kernel",
        "tile_order_kernel", "zero_rows_kernel", "stop_depth_bounds_kernel", "region_bin_kernel", "region_prepare_kernel",
        "status_tag_kernel", "rs_small_sort_kernel", "step_uninstanced_kernel", "chain_kernel",
        "tb_entries_kernel", "tb_emit_kernel", "tb_regions_from_counts_kernel", "tb_regions_kernel", "tb_tile_count_kernel",
        "tb_region_scan_kernel", "tb_tile_scan_kernel", "tb_write_kernel", "depth_l1_kernel", "rows_pack_kernel", "row_mask_kernel")


def short(name):
    for o in sorted(OURS, key=len, reverse=True):   # longest first: dwt2_l1_fwd_kernel contains l1_fwd_kernel
        if o in name:
            return o
    return name.split("(")[0][-60:]


stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
import shutil
shutil.copyfile(stats, os.path.join(here, "%s_rocprofv3_kernel_stats_full.csv" % tag))  # the unedited summary
with open(os.path.join(here, "%s_kernel_stats.csv" % tag), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:40]:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
tr = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(tr) if "render_bwd_wave_kernel" in r["Kernel_Name"] or "render_bwd_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
# the bench ends with a few steps in its one-launch form (GS_TWO_PHASE_STEP=0, for the per-kernel rooflines): take the last
# step that ran the two-phase form when there is one
for j in range(len(idx) - 2, 0, -1):
    if any("step_uninstanced_kernel" in r["Kernel_Name"] for r in tr[idx[j - 1]:idx[j]]):
        a, b = idx[j - 1], idx[j]
        break
seg = tr[a:b]
wall = (int(tr[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
# kernels of the two streams overlap: GPU-busy = length of the union of the intervals
t_end = int(tr[b]["Start_Timestamp"])   # (the side stream's kernel of the NEXT step starts just before that step's render_bwd)
iv = sorted((int(r["Start_Timestamp"]), min(int(r["End_Timestamp"]), t_end)) for r in seg)
busy, cur_s, cur_e = 0.0, None, None
for s0, e0 in iv:
    if cur_e is None or s0 > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s0, e0
    else:
        cur_e = max(cur_e, e0)
busy = (busy + (cur_e - cur_s if cur_e is not None else 0)) / 1e6
agg = collections.OrderedDict()
for r in seg:
    n = short(r["Kernel_Name"])
    agg.setdefault(n, [0.0, 0])
    agg[n][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[n][1] += 1
with open(os.path.join(here, "%s_step_timeline.txt" % tag), "w") as f:
    f.write("one steady-state train step (render_bwd to render_bwd), rocprofv3 --kernel-trace\n")
    f.write("wall %.3f ms, GPU busy %.3f ms, %d kernel launches\n" % (wall, busy, len(seg)))
    for n, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0]):
        f.write("%10.1f us %4d  %s\n" % (d, c, n))

traffic = {}
table = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for kind in ("fetch", "write"):
    files = glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv"))
    if not files:
        continue
    for r in csv.DictReader(open(files[0])):
        table[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(here, "%s_pmc_traffic.csv" % tag), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "hbm_read_bytes_corrected(x2)",
                "hbm_write_bytes", "hbm_total_bytes_per_launch"])
    for k, v in sorted(table.items()):
        if k not in OURS or not v["FETCH_SIZE"]:
            continue
        # steady-state launches only: drop the setup launches (GT renders etc.) by taking the last third
        fs = v["FETCH_SIZE"][-max(1, len(v["FETCH_SIZE"]) // 3):]
        ws = v["WRITE_SIZE"][-max(1, len(v["WRITE_SIZE"]) // 3):] if v["WRITE_SIZE"] else [0.0]
        fm, wm = sum(fs) / len(fs), sum(ws) / len(ws)
        rd, wr = 2.0 * fm * 1024.0, wm * 1024.0
        w.writerow([k, len(v["FETCH_SIZE"]), "%.1f" % fm, "%.1f" % wm, "%.0f" % rd, "%.0f" % wr, "%.0f" % (rd + wr)])
        traffic[k] = rd + wr
stage_of = {"render_bwd": "render_bwd_wave_kernel", "render_fwd": "render_fwd_wave_kernel", "preprocess_fwd": "preprocess_fwd_kernel",
            "preprocess_bwd": "preprocess_bwd_kernel", "preprocess_bwd_step": "preprocess_bwd_step_kernel",
            "duplicate": "duplicate_kernel", "tile_ranges": "tile_ranges_kernel", "region_bin": "region_bin_kernel",
            "ssim_fwd": "ssim_fwd_kernel", "ssim_bwd": "ssim_bwd_kernel",
            "step_uninstanced": "step_uninstanced_kernel", "chain": "chain_kernel"}
out = {st: traffic[k] for st, k in stage_of.items() if k in traffic}
out.update(run_identity())
out["launches_counted"] = {st: len(table[k]["FETCH_SIZE"]) for st, k in stage_of.items() if k in traffic}
if "region_bin_kernel" in traffic:   # region binning: the whole binning stage is this one kernel (bench.py stage "sort")
    out["sort"] = traffic["region_bin_kernel"]
elif "rs_scatter_kernel" in traffic:  # LSD path: all passes of hist + scatter (launch counts per step: 6 each)
    out["sort"] = 6 * (traffic.get("rs_scatter_kernel", 0) + traffic.get("rs_hist_kernel", 0))
json.dump(out, open(os.path.join(here, "pmc_traffic.json"), "w"), indent=1)
print(open(os.path.join(here, "%s_step_timeline.txt" % tag)).read())
print(open(os.path.join(here, "%s_pmc_traffic.csv" % tag)).read())

# SQ counters -> <tag>_sq_counters.csv (per kernel: instructions per wave, VALU-busy share of the SIMD cycles, wait shares)
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "sq[0-9]*", "*", "*_counter_collection.csv")):   # (not sq_<scene>: the other scenes' passes)
    for r in csv.DictReader(open(f)):
        sq[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
if sq:
    with open(os.path.join(here, "%s_sq_counters.csv" % tag), "w") as f:
        w = csv.writer(f)
        # VALU_issue_pct_of_peak: wave-instructions x 2 cycles (the wave64 issue rate a SIMD sustains with >= 2 ready waves)
        # over all SIMD cycles of the launch; VALU_active_x4: SQ_ACTIVE_INST_VALU (quad-cycles) x 4, which prices every
        # instruction at the 4-cycle single-wave issue cost and so reads high (round 1 quoted it as "busy")
        w.writerow(["kernel", "waves", "VALU_insts_per_wave", "LDS_insts_per_wave", "SALU_insts_per_wave", "VMEM_rd_per_wave",
                    "VMEM_wr_per_wave", "VALU_issue_pct_of_peak", "VALU_active_x4_pct_of_SIMD_cycles", "wait_any_pct_of_wave_cycles",
                    "wait_inst_pct_of_wave_cycles", "waves_per_SIMD", "LDS_bank_conflict_pct_of_LDS_cycles"])
        sq_insts = run_identity()
        sq_insts["launches_counted"] = {}
        for k, v in sorted(sq.items()):
            m = {c: sum(x) / len(x) for c, x in v.items()}
            if k not in OURS or "GRBM_GUI_ACTIVE" not in m or m.get("SQ_WAVES", 0) == 0:
                continue
            simd_cycles = m["GRBM_GUI_ACTIVE"] / 8 * 1024          # 8 XCDs counted; 256 CUs x 4 SIMDs
            wc = max(1.0, m.get("SQ_WAVE_CYCLES", 0))
            w.writerow([k, int(m["SQ_WAVES"]), round(m.get("SQ_INSTS_VALU", 0) / m["SQ_WAVES"], 1),
                        round(m.get("SQ_INSTS_LDS", 0) / m["SQ_WAVES"], 1), round(m.get("SQ_INSTS_SALU", 0) / m["SQ_WAVES"], 1),
                        round(m.get("SQ_INSTS_VMEM_RD", 0) / m["SQ_WAVES"], 1), round(m.get("SQ_INSTS_VMEM_WR", 0) / m["SQ_WAVES"], 1),
                        round(100 * m.get("SQ_INSTS_VALU", 0) * 2 / simd_cycles, 1),
                        round(100 * m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / simd_cycles, 1),   # quad-cycles -> cycles
                        round(100 * m.get("SQ_WAIT_ANY", 0) / wc, 1), round(100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 1),
                        round(wc * 4 / simd_cycles, 2),
                        round(100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 0)), 1)])
            for st, kn in (("render_bwd", "render_bwd_wave_kernel"), ("render_fwd", "render_fwd_wave_kernel")):
                if k == kn:
                    sq_insts[st] = m.get("SQ_INSTS_VALU", 0)   # wave-instructions per launch (mean over the launches)
                    sq_insts["launches_counted"][st] = len(v.get("SQ_INSTS_VALU", []))
                    # SIMD cycles the VALU spent issuing this kernel's instructions (SQ_ACTIVE_INST_VALU counts quad-cycles)
                    sq_insts[st + "_valu_active_cycles_per_simd"] = m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024
                    sq_insts[st + "_valu_active_share_under_profiler"] = m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / simd_cycles
    json.dump(sq_insts, open(os.path.join(here, "sq_insts.json"), "w"), indent=1)
    print(open(os.path.join(here, "%s_sq_counters.csv" % tag)).read())


# SURVEY 8(d)'s other inputs (collect.sh: GS_BENCH_SCENE=<kind> passes) -> <tag>_<kind>_kernel_stats.csv, sq_insts_<kind>.json
for kind in ("init_like",):
    st_files = glob.glob(os.path.join(src, "trace_" + kind, "*", "*_kernel_stats.csv"))
    if not st_files:
        continue
    shutil.copyfile(st_files[0], os.path.join(here, "%s_%s_rocprofv3_kernel_stats_full.csv" % (tag, kind)))
    with open(os.path.join(here, "%s_%s_kernel_stats.csv" % (tag, kind)), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in list(csv.DictReader(open(st_files[0])))[:30]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    ident = {"tag": tag, "source_id": source_id(os.path.dirname(script_dir)), "scene": kind}
    try:
        line = [x for x in open(os.path.join(src, "bench_trace_%s.json" % kind)).read().splitlines() if x.startswith("{")][-1]
        j = json.loads(line)
        ident["P"], ident["R"] = j["config"]["gaussians"], j["config"]["num_rendered_last_view"]
        shutil.copyfile(os.path.join(src, "bench_trace_%s.json" % kind), os.path.join(here, "%s_%s_bench_under_rocprof.json" % (tag, kind)))
    except Exception as e:   # noqa: BLE001
        ident["identity_error"] = repr(e)
    sqk = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "sq_" + kind, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            sqk[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ident["launches_counted"] = {}
    for stn, kn in (("render_bwd", "render_bwd_wave_kernel"), ("render_fwd", "render_fwd_wave_kernel")):
        v = sqk.get(kn, {}).get("SQ_INSTS_VALU", [])
        if v:
            ident[stn] = sum(v) / len(v)
            ident["launches_counted"][stn] = len(v)
            wv = sqk[kn].get("SQ_WAVES", [])
            sa = sqk[kn].get("SQ_INSTS_SALU", [])
            if wv and sum(wv):
                ident[stn + "_valu_per_wave"] = sum(v) / sum(wv) * len(wv) / len(v)
                if sa:
                    ident[stn + "_salu_per_wave"] = sum(sa) / sum(wv) * len(wv) / len(sa)
    json.dump(ident, open(os.path.join(here, "sq_insts_%s.json" % kind), "w"), indent=1)
    print(kind, json.dumps(ident))


# the two-level binning of the reference's lists (collect.sh: tests/tools/binning_probe.py c3 0) -> <tag>_binning_kernel_stats.csv
# (kernel trace summary of the forwards) and <tag>_binning_counters.csv (bytes per launch, LDS bank conflicts)
st_files = glob.glob(os.path.join(src, "trace_binning", "*", "*_kernel_stats.csv"))
if st_files:
    BIN = ("tb_", "rs_", "scan_sums_kernel", "scan_block_sums_kernel", "bin_prepare_kernel", "preprocess_fwd_kernel", "render_fwd_wave_kernel")
    with open(os.path.join(here, "%s_binning_kernel_stats.csv" % tag), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for r in csv.DictReader(open(st_files[0])):
            if any(b in r["Name"] for b in BIN):
                w.writerow([r["Name"].split("(")[0][-48:], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
    try:
        shutil.copyfile(os.path.join(src, "binning_trace.log"), os.path.join(here, "%s_binning_probe.log" % tag))
    except OSError:
        pass
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for d_ in ("sq_binning", "fetch_binning", "write_binning"):
        for f in glob.glob(os.path.join(src, d_, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                cnt[r["Kernel_Name"].split("(")[0][-48:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(here, "%s_binning_counters.csv" % tag), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "hbm_read_bytes_corrected(x2)", "hbm_write_bytes", "LDS_insts_per_wave",
                    "LDS_bank_conflict_pct_of_LDS_cycles"])
        for k, v in sorted(cnt.items()):
            if not any(b in k for b in BIN):
                continue
            last = lambda x: x[-max(1, len(x) // 2):]    # (the steady-state launches: the later half)
            m = {c: sum(last(x)) / len(last(x)) for c, x in v.items()}
            w.writerow([k, len(next(iter(v.values()))), "%.0f" % (2048.0 * m.get("FETCH_SIZE", 0)), "%.0f" % (1024.0 * m.get("WRITE_SIZE", 0)),
                        round(m.get("SQ_INSTS_LDS", 0) / max(1.0, m.get("SQ_WAVES", 0)), 1),
                        round(100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 0)), 1)])
    print(open(os.path.join(here, "%s_binning_counters.csv" % tag)).read())
