#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh on the MI355X box) into the committed summaries
under profiles/:

    <tag>_bench.json, <tag>_bench_720p.json, <tag>_bench_pipeline.json      the bench lines
    <tag>_bench_kernel_stats.csv, <tag>_bench_720p_kernel_stats.csv,
    <tag>_ba_kernel_stats.csv                                               rocprofv3 --kernel-trace --stats
    <tag>_hbm_traffic.json        FETCH_SIZE / WRITE_SIZE per launch and kernel (+ the hash of the kernel source)
    <tag>_fast_issue.json         SQ instruction-issue counters of the shipped FAST kernel, VGA x 64 and 720p x 64
    <tag>_ba_mfma.json            matrix-core counters of the local-BA kernels per grid shape (single window / batch)

    python tools/summarize_profiles.py r02 [frames_per_launch]

FETCH_SIZE / WRITE_SIZE come from separate --pmc passes (they do not fit one pass) and are averaged per launch and
kernel.  Units: rocprofv3 reports KB (bytes = value * 1024).  gfx950 caveat (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE halves wide 16-B/lane streaming reads; the ORB kernels read 1-4 B per lane, an uncalibrated width, so the
values are reported as measured and the comparison with the algorithmic byte count is indicative."""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("__amd_rocclr", "at::native", "Cijk", "ncclDevKernel")
WIDE_READERS = ("fast_cells_kernel", "describe_kernel", "ic_angle_kernel", "resize_rows8_kernel", "hamming_best2_mfma_kernel")   # 16 bytes per lane and load


def newest(path, pattern):
    f = glob.glob(os.path.join(path, "**", pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def kname(row):
    return row["Kernel_Name"].split("(")[0].replace("void ", "")


def per_kernel(path, counter):
    f = newest(path, "*counter_collection.csv")
    if not f:
        return {}
    acc = {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        d = acc.setdefault(kname(r), {})
        key = r.get("Dispatch_Id")
        d[key] = d.get(key, 0.0) + float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}


def counters_by_kernel_and_grid(path, want=None):
    """{(kernel, grid): {counter: mean per launch, 'launches': n}}"""
    f = newest(path, "*counter_collection.csv")
    if not f:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(dict))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = kname(r)
        if k.startswith(SKIP) or (want and not any(w in k for w in want)):
            continue
        key = (k, int(r["Grid_Size"]))
        d = acc[key][r["Counter_Name"]]
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        dur[key][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out = {}
    for key, cs in acc.items():
        out[key] = {c: sum(v.values()) / len(v) for c, v in cs.items()}
        out[key]["launches"] = max(len(v) for v in cs.values())
        out[key]["avg_us_in_counter_pass"] = sum(dur[key].values()) / len(dur[key])
    return out


def kernel_avg_us(path, name):
    f = newest(path, "*kernel_stats.csv")
    if not f:
        return None
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith(name) or r["Name"].replace("void ", "").startswith(name):
            return float(r["AverageNs"]) / 1e3
    return None


def sha16(path):
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def copy_bench(src, name, dst):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return False
    lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
    if not lines:
        return False
    json.dump(json.loads(lines[-1]), open(dst, "w"), indent=1)
    return True


# Vector instructions per wave (= per FAST cell) that THIS formulation of exact FAST cannot go below, phase by phase, next to what
# the shipped kernel issues (static counts of the phases' basic blocks in the gfx950 ISA x the trip counts of the bench content:
# DESIGN.md section 5 has the table).  The sum, per pyramid pixel and lane, is what bench.py reports as floor_lane_ops_per_px.
FAST_BUDGET = [
    # phase, issued now, floor, why the floor
    ("stage the window (LDS-DMA) + zero the score tile + cell set-up", 74, 30, "one address per 16-byte chunk and the cell's constants"),
    ("pre-test, 4 pixels per lane and step (4.2 steps per cell)", 4.2 * 27, 4.2 * 21, "2 v_alignbyte + v_not + 4 difference lerps + 8 threshold lerps + 4 combines + 2 mask merges"),
    ("survivor append: count, wave prefix sum, 4 slot stores (4.2 steps)", 4.2 * 27, 4.2 * 16, "popcount 3 + DPP scan 7 + address 2 + one add per slot"),
    ("exact score, 2 entries per lane (262 entries per cell: 2.55 passes issued, 2.05 without the half-empty last pass)", 2.55 * 112, 2.05 * 82, "34 half-word packs + 40 v_pk_min3 / max3 + 8 for score, compare and store"),
    ("non-maximum suppression over the listed corners (118 per cell)", 37, 20, "8 neighbour reads and a 3-level max per corner, two 64-lane rounds"),
    ("candidate output (25 per cell)", 32, 20, "decode, score fetch, one 8-byte store per candidate"),
]


def fast_floor(vga):
    c = vga["counters_per_launch"]
    per_wave = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    issued = sum(b[1] for b in FAST_BUDGET)
    floor = sum(b[2] for b in FAST_BUDGET)
    lane_ops_now = 64.0 * c["SQ_INSTS_VALU"] / vga["pyramid_pixels_per_launch"]
    return {
        "lane_ops_per_px": round(lane_ops_now * floor / per_wave, 1),
        "valu_per_cell_measured": round(per_wave, 1), "valu_per_cell_accounted": round(issued, 1), "valu_per_cell_floor": round(floor, 1),
        "phases": [{"phase": b[0], "issued": round(b[1], 1), "floor": round(b[2], 1), "why": b[3]} for b in FAST_BUDGET],
        "derivation": "exact FAST-9/16 per 30 x 30 cell on this content (27.9 %% of the scan pixels pass the compass pre-test, 12.5 %% are corners): "
                      "per cell %.0f vector instructions issued (SQ_INSTS_VALU / SQ_WAVES; %.0f accounted by phase), %.0f if every phase ran at the "
                      "instruction count its arithmetic needs (phases: profiles/<tag>_fast_issue.json floor.phases, DESIGN.md section 5); "
                      "floor = measured lane-ops per pixel x floor / issued.  The kernel is at %.2f x this floor; the floor itself is %.1f x the "
                      "5.6 lane-ops per pixel that 0.70 of the HBM roofline would allow." % (
                          per_wave, issued, floor, per_wave / floor, lane_ops_now * floor / per_wave / 5.6),
    }


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if not frames:   # frames per launch = the bench's default batch, as the collected bench line states it
        try:
            line = [l for l in open(os.path.join(src, "bench.json")).read().splitlines() if l.startswith("{")][-1]
            frames = int(json.loads(line)["config"]["frames_per_step_per_gpu"])
        except Exception:
            frames = 128
    dst = os.path.join(ROOT, "profiles")
    done = []
    for sub, out in (("trace", "_bench_kernel_stats.csv"), ("trace720", "_bench_720p_kernel_stats.csv"), ("ba_trace", "_ba_kernel_stats.csv")):
        f = newest(os.path.join(src, sub), "*kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(dst, tag + out)); done.append(out)
    for name, out in (("bench.json", "_bench.json"), ("bench_720p.json", "_bench_720p.json"), ("bench_pipeline.json", "_bench_pipeline.json")):
        if copy_bench(src, name, os.path.join(dst, tag + out)):
            done.append(out)
    # ---- HBM traffic ----
    fetch = per_kernel(os.path.join(src, "fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(src, "write"), "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith(SKIP):
            continue
        kernels[k] = {"FETCH_SIZE_KB_per_launch": round(fetch.get(k, 0.0), 1), "WRITE_SIZE_KB_per_launch": round(write.get(k, 0.0), 1)}
        # gfx950: FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes (MI355X_MICROARCH.md, HBM): the kernels whose
        # reads are of that kind (LDS-DMA / dwordx4 window loads) get the doubled figure next to the raw one
        if k.startswith(WIDE_READERS):
            kernels[k]["FETCH_KB_corrected_x2"] = round(2 * fetch.get(k, 0.0), 1)
    if kernels:
        doc = {"_about": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 bench.py --no-ba --no-cpu --steps 3 "
                         "--warmup 1` on MI355X, averaged per launch; %d VGA frames per launch. Units: KB as rocprofv3 reports them "
                         "(bytes = value*1024). gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of the bytes of "
                         "16-B/lane reads; kernels that read that way carry FETCH_KB_corrected_x2 (the others read 4-12 B per lane, an "
                         "uncalibrated width: as measured)." % frames,
               "frames_per_launch": frames,
               "kernel_src_sha16": sha16(os.path.join(ROOT, "weiner_slamit_v2_amd", "csrc", "orb_kernels.hip")),
               "kernels": kernels}
        json.dump(doc, open(os.path.join(dst, tag + "_hbm_traffic.json"), "w"), indent=1); done.append("_hbm_traffic.json")
    # ---- FAST issue counters ----
    geo = {"vga": (640 * 480, 950532, "trace", "bench.json"), "720p": (1280 * 720, 2853088, "trace720", "bench_720p.json")}
    fast = {}
    frames_vga = frames
    for cfg, (_, pyr_px, tr, bj) in geo.items():
        frames = frames_vga
        try:   # frames per launch of this geometry's runs = the default batch of its bench line
            line = [l for l in open(os.path.join(src, bj)).read().splitlines() if l.startswith("{")][-1]
            frames = int(json.loads(line)["config"]["frames_per_step_per_gpu"])
        except Exception:
            pass
        c = {}
        for part in ("sq_a_", "sq_b_"):
            for (k, grid), v in counters_by_kernel_and_grid(os.path.join(src, part + cfg), want=("fast_cells_kernel",)).items():
                c.update({kk: vv for kk, vv in v.items() if kk not in ("launches", "avg_us_in_counter_pass")})
        if not c:
            continue
        us = kernel_avg_us(os.path.join(src, tr), "fast_cells_kernel")
        px = pyr_px * frames
        e = {"frames_per_launch": frames, "pyramid_pixels_per_launch": px, "counters_per_launch": {k: round(v) for k, v in sorted(c.items())},
             "avg_launch_us_rocprof": round(us, 2) if us else None}
        if "SQ_INSTS_VALU" in c:
            total = c.get("SQ_INSTS_VALU", 0) + c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_LDS", 0) + c.get("SQ_INSTS_VMEM_RD", 0)
            e["valu_wave_instructions_per_pixel"] = round(c["SQ_INSTS_VALU"] / px, 4)
            e["valu_lane_ops_per_pixel"] = round(64 * c["SQ_INSTS_VALU"] / px, 2)
            e["all_wave_instructions_per_pixel"] = round(total / px, 4)
            if us:
                e["valu_instructions_per_simd_per_us"] = round(c["SQ_INSTS_VALU"] / 1024 / us, 1)
                e["ns_per_valu_instruction_per_simd"] = round(1e3 * us * 1024 / c["SQ_INSTS_VALU"], 3)
                e["achieved_GBps_algorithmic"] = round(px / us / 1e3, 1)
                e["frac_of_8TBps"] = round(px / us / 1e3 / 8000.0, 4)
        fast[cfg] = e
    if fast:
        doc = {"_about": "SQ counters of the shipped fast_cells_kernel (rocprofv3 --kernel-trace --pmc, two passes of 8 counters, `python3 "
                         "bench.py --no-ba --no-cpu --config <cfg> --steps 3 --warmup 1`), means per launch (frames_per_launch: the configuration's default batch), with the launch duration of "
                         "the --kernel-trace --stats run of the same build.  1024 SIMDs.  Measured issue cost on this chip "
                         "(tools/diag/ubench, wall clock at 4 waves per SIMD): 2.29 ns per half-rate VALU wave-instruction per SIMD "
                         "(v_lerp_u8, v_perm, v_pk_*, 3-operand integer ops), 1.53 ns per full-rate one (v_and/or/xor/add, v_bitop3).",
               "kernel_src_sha16": sha16(os.path.join(ROOT, "weiner_slamit_v2_amd", "csrc", "orb_kernels.hip")), "geometries": fast}
        frames = frames_vga
        if "vga" in fast and "SQ_INSTS_VALU" in fast["vga"]["counters_per_launch"] and "SQ_WAVES" in fast["vga"]["counters_per_launch"]:
            doc["floor"] = fast_floor(fast["vga"])
        json.dump(doc, open(os.path.join(dst, tag + "_fast_issue.json"), "w"), indent=1); done.append("_fast_issue.json")
    # ---- BA matrix-core counters per kernel and grid shape ----
    ba = counters_by_kernel_and_grid(os.path.join(src, "ba_mfma"), want=("k_schur", "k_ldlt_solve"))
    if ba:
        rows = []
        for (k, grid), v in sorted(ba.items()):
            us = v.pop("avg_us_in_counter_pass")
            r = {"kernel": k, "grid_threads": grid, "launches": v.pop("launches"), "avg_us_in_counter_pass": round(us, 2)}
            r.update({kk: round(vv) for kk, vv in sorted(v.items())})
            # matrix-pipe busy cycles summed over the SIMDs / (1024 SIMDs x launch duration at the 2.4 GHz nominal clock)
            r["mfma_busy_frac_of_chip"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * us * 2400.0), 5)
            r["executed_TFLOPs"] = round(v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512 / us / 1e6, 3)
            rows.append(r)
        doc = {"_about": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES "
                         "SQ_WAVES -- python3 tools/bench_ba.py 2 (window-8 and dense 50 KF x 2000 pts, single windows and batches of 8), means per "
                         "launch, split by grid size (threads): the small grids are single windows.  SQ_INSTS_VALU_MFMA_MOPS_F64 counts in units of "
                         "512 flop; one v_mfma_f64_16x16x4_f64 is 2048 flop.  mfma_busy_frac_of_chip = SQ_VALU_MFMA_BUSY_CYCLES (summed over SIMDs) / "
                         "(1024 SIMDs x duration x 2.4 GHz); executed_TFLOPs = MOPS x 512 / duration (fp64 MFMA peak 78.6).  A launch mixes window-8 "
                         "and dense windows (tools/bench_ba.py runs both); SURVEY 8(d)'s ALGORITHMIC Schur flops per LM trial are 17.9 MFLOP "
                         "(window-8) and 565 MFLOP (dense) per window.",
               "rows": rows}
        json.dump(doc, open(os.path.join(dst, tag + "_ba_mfma.json"), "w"), indent=1); done.append("_ba_mfma.json")
    print("profiles/%s: %s" % (tag, " ".join(done)))


if __name__ == "__main__":
    main()
