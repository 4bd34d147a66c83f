#!/usr/bin/env python3
"""bench.py — the hot path's headline benchmark on MI355X (BASELINE.json).

Metric: frames/sec of ORB extract + all-pairs Hamming match at 640x480, 8 levels, 1000 features
(BASELINE.json configs[1]); secondary: local-BA LM iterations/sec (50 KF, 2000 points, configs[3]).

A "step" = one batch of B independent camera streams advancing by one frame: the B frames
(resident in HBM before the timed region) go through the whole extractor (pyramid, FAST cells,
octree, orientation, blur, rBRIEF) and each frame's descriptors are matched against the same
stream's previous frame (best / second best / TH_LOW + ratio test inputs).  Streams are
independent, so with N GPUs each rank owns B streams (weak scaling) and only a per-frame result
summary is all-gathered over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --config pipeline [--streams 8]      # BASELINE configs[4]: 8 x 720p streams + per-GPU local BA + slot gather

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import platform
import re
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NFEAT, NLEVELS = 640, 480, 1000, 8
FAST_BYTES_PER_FRAME = 950532        # 1 B per pyramid pixel, 8 levels of 640x480
CONFIGS = {  # name: (width, height, nfeatures, algorithmic FAST bytes per frame, BASELINE.json config)
    "vga": (640, 480, 1000, 950532, "configs[1]"),
    "720p": (1280, 720, 2000, 2853088, "configs[2]"),
    "pipeline": (1280, 720, 2000, 2853088, "configs[4]"),
}
# algorithmic flops of the Schur complement per LM trial (50 KF x 2000 points, mono edges)
SCHUR_ALGO_MFLOP = {"window8": 17.9, "dense": 565.0}
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6
# Integer vector peak of the chip: 1024 SIMDs x 16 lanes per clock x 2.4 GHz (full-rate instructions; the byte instructions
# FAST lives on -- v_lerp_u8, v_perm, v_pk_* -- run at half of it)
VALU_PEAK_TLANEOPS = 39.3


def _sha16(path):
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def measured_traffic(batch):
    """HBM bytes per launch of the FAST kernel from the newest committed rocprofv3 PMC summary
    (profiles/rNN_hbm_traffic.json, tools/collect_profiles.sh + tools/summarize_profiles.py): FETCH_SIZE + WRITE_SIZE,
    scaled to this run's frames per launch.  (None, reason) when there is no summary or when it was taken on a different
    version of the kernel source (the summary records the source's hash): a stale number is worse than none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None, "no profiles/r*_hbm_traffic.json"
    path = files[-1]
    try:
        doc = json.load(open(path))
        k = [v for n, v in doc["kernels"].items() if n.startswith("fast_cells_kernel")][0]
        kb = k.get("FETCH_KB_corrected_x2", k["FETCH_SIZE_KB_per_launch"]) + k["WRITE_SIZE_KB_per_launch"]   # (16-B/lane reads: FETCH_SIZE x 2 on gfx950)
        rel = os.path.relpath(path, ROOT)
        sha = _sha16(os.path.join(ROOT, "weiner_slamit_v2_amd", "csrc", "orb_kernels.hip"))
        if doc.get("kernel_src_sha16") != sha:
            return None, "%s was taken on another version of csrc/orb_kernels.hip (%s, now %s)" % (rel, doc.get("kernel_src_sha16"), sha)
        return int(kb * 1024 * batch / doc["frames_per_launch"]), "%s (rocprofv3 --pmc FETCH_SIZE x 2 [gfx950: 16-B/lane reads are tallied at half] + WRITE_SIZE, separate passes; bytes per launch; kernel source %s)" % (rel, sha)
    except Exception as e:
        return None, "unreadable %s: %r" % (path, e)


def measured_valu(batch, fast_avg_ms, pixels_per_frame):
    """The FAST kernel against the chip's VECTOR-instruction roofline: lane-operations per pyramid pixel from the newest
    committed SQ counter summary (profiles/rNN_fast_issue.json: SQ_INSTS_VALU per launch of the shipped kernel), priced with
    THIS run's launch time.  None when the summary was taken on another version of the kernel source."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fast_issue.json")))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
        sha = _sha16(os.path.join(ROOT, "weiner_slamit_v2_amd", "csrc", "orb_kernels.hip"))
        rel = os.path.relpath(files[-1], ROOT)
        if doc.get("kernel_src_sha16") != sha:
            return {"lane_ops_per_px": None, "source": "%s was taken on another version of csrc/orb_kernels.hip (%s, now %s)" % (rel, doc.get("kernel_src_sha16"), sha)}
        k = doc["geometries"]["vga"]
        valu_per_frame = k["counters_per_launch"]["SQ_INSTS_VALU"] / k["frames_per_launch"]
        lane_ops_px = 64.0 * valu_per_frame / pixels_per_frame
        achieved = 64.0 * valu_per_frame * batch / (fast_avg_ms * 1e-3) / 1e12 if fast_avg_ms > 0 else 0.0
        out = {"lane_ops_per_px": round(lane_ops_px, 2), "valu_wave_instructions_per_frame": round(valu_per_frame),
               "achieved_Tlaneops": round(achieved, 2), "peak_Tlaneops": VALU_PEAK_TLANEOPS, "frac": round(achieved / VALU_PEAK_TLANEOPS, 4),
               "source": "%s (rocprofv3 --pmc SQ_INSTS_VALU on the shipped kernel, kernel source %s) x this run's launch time" % (rel, sha)}
        if "floor" in doc:
            out["floor_lane_ops_per_px"] = doc["floor"]["lane_ops_per_px"]
            out["floor_derivation"] = doc["floor"]["derivation"]
        return out
    except Exception as e:
        return {"lane_ops_per_px": None, "source": "unreadable %s: %r" % (files[-1], e)}


def host_api_latency(fa, fb):
    """SURVEY 8(d)'s latency figures through the HOST-POINTER C-ABI (pageable upload + kernels + download, what one Tracking
    thread sees): one VGA frame, a batch of 64 VGA frames, one 1000 x 1000 best/second match.  Medians, milliseconds."""
    from weiner_slamit_v2_amd import api

    def med(fn, n):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(1e3 * (time.perf_counter() - t0))
        return round(statistics.median(ts), 4)

    e1 = api.ORBextractor(1000, 1.2, NLEVELS, 20, 7, max_batch=1)
    e64 = api.ORBextractor(1000, 1.2, NLEVELS, 20, 7, max_batch=64)
    batch = np.stack([fa[i % len(fa)] for i in range(64)])
    _, da = e1(fa[0])
    _, db = e1(fb[0])
    da, db = np.ascontiguousarray(da[:1000]), np.ascontiguousarray(db[:1000])
    out = {"extract_1_vga_ms": med(lambda: e1(fa[0]), 20),
           "extract_batch64_ms": med(lambda: e64.extract_batch(batch), 5),
           "best2_1000x1000_ms": med(lambda: api.ORBmatcher.best2(db, da), 20),
           "note": "host-pointer C-ABI incl. upload and download (slamit_orb_extract, slamit_orb_extract_batch, slamit_hamming_best2); medians"}
    del e1, e64
    return out


def roofline_720p(dev, steps=5):
    """The FAST pass on the geometry north_star's >= 70 % bar is stated on: 64 frames of 1280x720, 2000 features, per launch
    (BASELINE configs[2]); `steps` extract + match steps after 2 warm-ups, the launch timed with HIP events like the headline's."""
    import torch
    from weiner_slamit_v2_amd import api, synth

    w, h, nfeat, bytes_per_frame, cfg = CONFIGS["720p"]
    B, uniq = 64, 8
    fa = [synth.synth_frame(w, h, 7000 + i) for i in range(uniq)]
    fb = [synth.warp_frame(fa[i], 7000 + i) for i in range(uniq)]
    d_frames = [torch.from_numpy(np.stack([f[i % uniq] for i in range(B)])).to(dev) for f in (fa, fb)]
    ext = api.ORBextractor(nfeat, 1.2, NLEVELS, 20, 7, device=dev.index, max_batch=B)
    ext._bind(w, h, B)
    cap = ext.max_keypoints
    d_kps = [torch.zeros((B, cap, 7), dtype=torch.float32, device=dev) for _ in range(2)]
    d_desc = [torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_n = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
    d_idx, d_best, d_second = (torch.zeros((B, cap), dtype=torch.int32, device=dev) for _ in range(3))
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step(k):
        ext.extract_batch_dev(d_frames[k & 1], d_kps[k & 1], d_desc[k & 1], d_n[k & 1], stream=stream)
        api.ORBmatcher.best2_batch_dev(d_desc[k & 1], d_n[k & 1], d_desc[(k - 1) & 1], d_n[(k - 1) & 1], d_idx, d_best, d_second,
                                       cap, device=dev.index, stream=stream)
    for k in range(2):
        step(k)
    torch.cuda.synchronize(dev)
    ext.profile(2)   # events around every launch of the FAST kernel
    t0 = time.perf_counter()
    for k in range(2, 2 + steps):
        step(k)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    fast_ms, calls = ext.profile(0)["fast"]
    avg = fast_ms / max(calls, 1)
    achieved = bytes_per_frame * B / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
    assert (d_n[(steps + 1) & 1].cpu().numpy() >= nfeat).all()
    return {"workload": "BASELINE %s: 64 frames of 1280x720 per launch, 2000 features" % cfg, "kernel": "fast_cells_kernel", "bound": "hbm",
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "avg_launch_ms": round(avg, 5), "launches": calls, "algorithmic_bytes_per_launch": bytes_per_frame * B, "traffic": None,
            "frames_per_s": round(B * steps / el, 1), "ms_per_step": round(1e3 * el / steps, 4),
            "note": "events around every launch drain the pipeline: frames_per_s here is a lower bound (bench.py --config 720p times the step)"}


def cpu_info():
    """What the CPU comparators ran on and how they were built (SURVEY 8d asks for both)."""
    model = platform.processor() or ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    flags = ""
    try:
        mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
        flags = re.search(r"^NATIVE_CXXFLAGS \?= (.*)$", mk, re.M).group(1)
    except Exception:
        pass
    try:
        cxx = subprocess.check_output(["g++", "--version"], text=True).splitlines()[0]
    except Exception:
        cxx = "g++ (unknown)"
    return {"model": model, "logical_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "compiler": cxx, "flags": flags}


def cpu_baseline(frames_a, frames_b, nfeat=1000, budget_s=12.0, min_frames=24):
    """The CPU port of the reference path on ONE host core over a bounded sample of the same workload: extract frame
    A_i, extract B_i, match B_i against A_i.  Timed on the comparator build of the oracle (-O3 -march=native, compiled on
    the machine it runs on; the strict -O2 build stays the parity checker) after 3 warm-up frames; `value` is frames per
    second over the median per-frame time of >= 20 timed frames."""
    from oracle import bindings as ob

    native = ob.use_native_comparator()
    orc = ob.OrbOracle(nfeat, 1.2, NLEVELS, 20, 7)
    times, prev, i = [], None, 0
    t_start = time.perf_counter()
    while True:
        src = frames_a if (i // len(frames_a)) % 2 == 0 else frames_b
        t0 = time.perf_counter()
        _, d = orc.extract(src[i % len(frames_a)])
        if prev is not None:
            ob.best2(d, prev)
        dt = time.perf_counter() - t0
        prev = d
        if i >= 3:
            times.append(dt)
        i += 1
        if time.perf_counter() - t_start >= budget_s and len(times) >= min_frames:
            break
    med = statistics.median(times)
    return {"value": round(1.0 / med, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "build": "oracle/liborb_oracle_native.so" if native else "oracle/liborb_oracle.so (-O2: the native comparator did not build)",
            "median_ms_per_frame": round(1e3 * med, 3), "mean_ms_per_frame": round(1e3 * sum(times) / len(times), 3),
            "sample": "%d synthetic %dx%d frames after 3 warm-ups, oracle extract (%d feat, 8 lvl) + all-pairs best2 vs previous frame, %.1f s" % (
                len(times), frames_a[0].shape[1], frames_a[0].shape[0], nfeat, time.perf_counter() - t_start)}


def cpu_baseline_all_cores(frames_a, frames_b, nfeat=1000, budget_s=8.0):
    """SURVEY 8(d)'s second comparator: one oracle instance per host thread (ctypes releases the GIL inside the
    oracle), every thread running the same extract + match-to-previous loop on its own stream."""
    import concurrent.futures as cf
    from oracle import bindings as ob

    ncores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    # a 1-GPU box owns a 16-core share of its host whatever the affinity mask says; SLAMIT_CPU_THREADS overrides
    ncores = int(os.environ.get("SLAMIT_CPU_THREADS", min(ncores, 16)))

    def worker(w):
        orc = ob.OrbOracle(nfeat, 1.2, NLEVELS, 20, 7)
        t0, n, prev, i = time.perf_counter(), 0, None, w
        while time.perf_counter() - t0 < budget_s:
            src = frames_a if (i // len(frames_a)) % 2 == 0 else frames_b
            _, d = orc.extract(src[i % len(frames_a)])
            if prev is not None:
                ob.best2(d, prev)
            prev, n, i = d, n + 1, i + 1
        return n, time.perf_counter() - t0

    ob.orb_lib()   # build / load once before the threads start
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(ncores) as ex:
        res = list(ex.map(worker, range(ncores)))
    el = time.perf_counter() - t0
    n = sum(r[0] for r in res)
    return {"value": round(n / el, 2), "unit": "frames/s", "cores": ncores, "kind": "port",
            "sample": "%d frames over %d threads (one oracle instance and stream per thread), %.1f s" % (n, ncores, el)}


def cpu_baseline_ba(prob):
    """The BA oracle (CPU port pinned to the reference's g2o) on one host core, one window."""
    from oracle import bindings as ob

    t0 = time.perf_counter()
    r = ob.ba_solve(prob)
    el = time.perf_counter() - t0
    return {"value": round(sum(r["stats"]["n_its"]) / el, 1), "unit": "iters/s", "cores": 1, "kind": "port",
            "sample": "1 window, %.0f ms" % (1e3 * el)}


def cpu_baseline_ba_reference(prob):
    """The reference's own g2o (oracle/_ref/libba_ref.so, built from /root/reference in the authoring container and
    shipped with the snapshot) on one host core, one window; None when the library is not there."""
    from oracle import bindings as ob

    if not ob.ba_ref_available():
        return None
    t0 = time.perf_counter()
    r = ob.ba_ref_solve(prob)
    el = time.perf_counter() - t0
    return {"value": round(sum(r["stats"]["n_its"]) / el, 1), "unit": "iters/s", "cores": 1, "kind": "reference",
            "sample": "1 window, %.0f ms (vendored g2o + Eigen of the reference)" % (1e3 * el)}


def ba_secondary(device, steps, with_cpu=True):
    """local-BA LM iterations/sec on BASELINE config 4 (50 KF x 2000 points): window-8 visibility
    (16,000 edges) and dense visibility (100,000 edges), single window and a batch of 8 windows,
    next to the CPU oracle (port of the reference's g2o path, pinned to it) on one host core."""
    try:
        from weiner_slamit_v2_amd import api, synth
        out = {"metric": "local-BA LM iterations/sec (50 KF, 2000 pts; schedule 5 robust + 10 plain)", "unit": "iters/s",
               "dtype": "f64", "note": "end to end through the host-pointer C-ABI (upload + solve + download); value = median of the single-window "
               "solves; batches hold DIFFERENT windows (seeds 12345 + i)"}
        for name, obs in (("window8", 8), ("dense", None)):
            # 64 DIFFERENT windows (seeds 12345 + i: same geometry, own noise and outliers), so that the windows of a batch
            # take different numbers of LM trials and do not run in lock step; the single-window figures use seed 12345
            probs = [synth.synth_ba(50, 2000, obs, seed=12345 + i) for i in range(64)]
            prob = probs[0]
            ne = max(len(q["edge_kf"]) for q in probs)
            opt = api.Optimizer(64, 2048, ne + 64, 8, device)
            opt.LocalBundleAdjustment(prob)  # warm-up
            reps = max(5, min(steps, 10))
            its, ts = 0, []
            for _ in range(reps):
                t0 = time.perf_counter()
                r = opt.LocalBundleAdjustment(prob)
                ts.append(time.perf_counter() - t0)
                its = sum(r["stats"]["n_its"])
            el = statistics.median(ts)
            opt.LocalBundleAdjustmentBatch(probs[:8])
            t1 = time.perf_counter()
            rb = opt.LocalBundleAdjustmentBatch(probs[:8])
            elb = time.perf_counter() - t1
            entry = {"edges": len(prob["edge_kf"]), "value": round(its / el, 1), "ms_per_window": round(1e3 * el, 3),
                     "ms_per_window_samples": [round(1e3 * t, 3) for t in ts],
                     "batch8_value": round(sum(sum(x["stats"]["n_its"]) for x in rb) / elb, 1)}
            # one profiled solve (events between the phases of every LM slot: slower, not part of the timings above)
            opt.profile(True)
            opt.LocalBundleAdjustment(prob)
            pr = opt.profile_read()
            opt.profile(False)
            schur_ms = pr["phase_ms"]["schur"] / max(pr["slots"], 1)
            algo = SCHUR_ALGO_MFLOP[name] * 1e6 / (schur_ms * 1e-3) / 1e12 if schur_ms > 0 else 0.0
            entry["phases_ms_per_slot"] = {k: round(v / max(pr["slots"], 1), 4) for k, v in pr["phase_ms"].items()}
            entry["slots"] = pr["slots"]
            entry["roofline"] = {"kernel": "k_schur_pose + k_schur_reduce (the Schur-complement phase of an LM slot)", "bound": "mfma",
                                 "achieved": round(algo, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(algo / FP64_MFMA_PEAK_TFLOPS, 5),
                                 "algorithmic_mflop_per_trial": SCHUR_ALGO_MFLOP[name], "executed_mflop_per_trial": round(pr["schur_exec_mflop"], 1),
                                 "executed_over_algorithmic": round(pr["schur_exec_mflop"] / SCHUR_ALGO_MFLOP[name], 2),
                                 "avg_phase_ms": round(schur_ms, 5), "traffic": None,
                                 "timing": "HIP events around the phase in a separate profiled solve (slamit_ba_profile), one window"}
            opt.close()
            opt = api.Optimizer(64, 2048, ne + 64, 64, device)   # also a batch of 64 windows per GPU
            opt.LocalBundleAdjustmentBatch(probs)
            t1 = time.perf_counter()
            rb = opt.LocalBundleAdjustmentBatch(probs)
            elb = time.perf_counter() - t1
            entry["batch64_value"] = round(sum(sum(x["stats"]["n_its"]) for x in rb) / elb, 1)
            trials = [sum(sum(x["stats"]["trials"][sg]) for sg in range(2)) for x in rb]
            entry["batch64_trials_min_max"] = [min(trials), max(trials)]
            if with_cpu:
                entry["cpu_baseline"] = cpu_baseline_ba(prob)
                entry["speedup_vs_cpu_1core"] = round(entry["value"] / entry["cpu_baseline"]["value"], 1)
                ref = cpu_baseline_ba_reference(prob)
                if ref:
                    entry["cpu_baseline_reference"] = ref
                    entry["speedup_vs_reference_g2o_1core"] = round(entry["value"] / ref["value"], 1)
            out[name] = entry
            opt.close()
        out["value"] = out["window8"]["value"]
        return out
    except Exception as e:  # the ORB line must still print
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5, help="the --steps-long timed region is repeated this many times, each bracketed by barrier + "
                    "synchronize; ms_per_step / value are the MEDIAN region's (SURVEY 8d), every region's figure is in repetitions_ms_per_step")
    # 256 streams per step: per-frame cost keeps falling with the batch (launch floor, ramp and tail of ~14 launches per step, the
    # octree's one long workgroup per frame, and the side stream's overlap all improve): 64 -> 218k frames/s, 128 -> 253k,
    # 256 -> 261k, 512 -> ~270k on one MI355X; 256 keeps a step at one millisecond
    ap.add_argument("--batch", type=int, default=None, help="independent camera streams per GPU and step (default: 256 at VGA, 64 at 720p "
                    "as BASELINE configs[2] words it)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--scene", choices=["rich", "sparse"], default="rich",
                    help="synthetic frame content (synth.synth_frame): 'rich' is the benchmark's workload (about one pixel in eight is a FAST corner); "
                         "'sparse' has a few percent of corners and shows how much the FAST pass depends on content (DESIGN.md section 5)")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline_720p and latency blocks (A/B and profiling runs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for real runs; gloo only to rehearse the multi-rank flow with ranks sharing one GPU")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="vga",
                    help="vga = the headline metric's configuration; 720p = BASELINE configs[2] (profiling run); "
                         "pipeline = BASELINE configs[4]: --streams 720p streams dealt over the ranks (stream s on rank s %% N), "
                         "each rank also solving its streams' local-BA windows, results gathered in fixed-capacity slots")
    ap.add_argument("--streams", type=int, default=8, help="pipeline config: total camera streams over all ranks")
    ap.add_argument("--ba-workers", type=int, default=6, help="pipeline config: local-BA batches in flight per rank (host threads, each with its own "
                    "BA handle and HIP stream); a step's slot carries the poses of the batch submitted that many steps earlier")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and run the collectives even on ONE rank (a one-GPU "
                    "test then drives RCCL: communicator, device-tensor all_gather_into_tensor, async work handles)")
    args = ap.parse_args()
    global W, H, NFEAT, FAST_BYTES_PER_FRAME
    W, H, NFEAT, FAST_BYTES_PER_FRAME, cfg_name = CONFIGS[args.config]
    pipeline = args.config == "pipeline"

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":   # rehearsal: every rank on GPU 0, summary tensors on the host
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        if args.force_dist:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            if args.backend == "gloo":
                dist.init_process_group("gloo", rank=0, world_size=1)
            else:
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    cdev = torch.device("cpu") if ((world > 1 or args.force_dist) and args.backend == "gloo") else dev   # where collective tensors live

    from weiner_slamit_v2_amd import api, shard, synth

    if pipeline:
        # strong scaling: a fixed set of streams, stream s on rank s % world; every stream has its own frames
        if args.streams % world:
            raise SystemExit("--streams must be a multiple of the number of ranks")
        mine = shard.stream_assignment(args.streams, world, rank)
        B = len(mine)
        uniq = B
        fa = [synth.synth_frame(W, H, 5000 + sid) for sid in mine]
        fb = [synth.warp_frame(fa[i], 5000 + sid) for i, sid in enumerate(mine)]
    else:
        B = args.batch if args.batch else (256 if args.config == "vga" else 64)
        # each rank owns its own B streams (different seeds per rank); two consecutive frames per stream; 16 distinct
        # frame pairs per rank, each used by B / 16 streams (the kernels are issue bound: content repeats do not help them)
        uniq = min(B, 16)
        fa = [synth.synth_frame(W, H, 1000 * rank + i, scene=args.scene) for i in range(uniq)]
        fb = [synth.warp_frame(fa[i], 1000 * rank + i) for i in range(uniq)]
    frames = [np.stack([f[i % uniq] for i in range(B)]) for f in (fa, fb)]
    d_frames = [torch.from_numpy(f).to(dev) for f in frames]

    ext = api.ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, device=dev.index, max_batch=B)
    ext._bind(W, H, B)
    cap = ext.max_keypoints
    # three result buffers in rotation: extract(k + 1) fills one while match(k) still reads the other two
    NBUF = 3
    d_kps = [torch.zeros((B, cap, 7), dtype=torch.float32, device=dev) for _ in range(NBUF)]
    d_desc = [torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
    d_n = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NBUF)]
    d_idx = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_best = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_second = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    gather = shard.SummaryGather(B, 2, cdev, world, force_collective=args.force_dist)  # per-frame (keypoints, matches) to every rank
    slots = ba = ba_probs = None
    ba_its = [0]
    if pipeline:
        # every stream also owns a local-BA window (BASELINE configs[3] geometry, window-8 visibility, its own noise and outliers:
        # seed 12345 + stream id); the rank's windows of a step are ONE batch, solved by one of --ba-workers host threads on its
        # own BA handle and HIP stream beside the extract + match launches.  Nothing in a step waits for the batch it submits:
        # the step's slots take the poses of the batch submitted --ba-workers steps earlier (shard.BaWorkers)
        slots = shard.SlotGather(B, cdev, world, force_collective=args.force_dist)
        ba_probs = [synth.synth_ba(50, 2000, 8, seed=12345 + sid) for sid in mine]
        ba_edges = max(len(p["edge_kf"]) for p in ba_probs) + 64
        ba = shard.BaWorkers(lambda: api.Optimizer(64, 2048, ba_edges, B, dev.index), args.ba_workers)
        for j in [ba.submit(ba_probs) for _ in range(ba.n)]:   # warm-up: every worker's handle, pinned block, kernels
            ba.result(j)
        # slot packing without per-step tensor construction: pinned staging for what the BA hands back on the host (iterations,
        # 50 poses per stream), constants on the device
        NPIN = 4   # a ring: the host runs ahead of the stream, so a staging buffer is reused only after the copy out of it has run
        pin_its = [torch.zeros(B, dtype=torch.int32).pin_memory() for _ in range(NPIN)]
        pin_pose = [torch.zeros((B, shard.SLOT_BA_KF, shard.SLOT_BA_DOUBLES), dtype=torch.float64).pin_memory() for _ in range(NPIN)]
        pin_ev = [torch.cuda.Event() for _ in range(NPIN)]
        last_ba = {"its": np.zeros(B, np.int32), "pose": np.zeros((B, shard.SLOT_BA_KF, shard.SLOT_BA_DOUBLES))}
        d_sid = torch.tensor(mine, dtype=torch.int32, device=dev)
        d_col = torch.arange(cap, device=dev, dtype=torch.int32)[None, :]
        ba_jobs = []
    # ONE non-default stream carries the whole step: extract(k) -> match(k) are ordered by the stream.  (A NULL
    # stream handle would mean "the extractor's own stream" to the C-ABI and un-order the two calls.)
    tstream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    # Opt-in experiment (SLAMIT_BENCH_TWO_STREAMS=1): the matcher of step k on a second stream, ordered after extract(k)
    # by an event, so that it overlaps the pyramid chain of extract(k + 1).  Measured 5 % SLOWER than one stream (0.369 vs
    # 0.352 ms per step): the two cross-stream events per step cost more than the overlap wins.  Default: one stream.
    two_streams = not pipeline and bool(os.environ.get("SLAMIT_BENCH_TWO_STREAMS"))
    mstream = torch.cuda.Stream(dev) if two_streams else tstream
    ev_x = [torch.cuda.Event() for _ in range(NBUF)]   # extract into buffer i finished
    ev_m = [torch.cuda.Event() for _ in range(NBUF)]   # the match that READ buffer i as "previous frame" finished

    def ba_take(res):
        """A finished batch becomes what the next slots carry (iterations + poses as quaternion | t); returns its LM iterations."""
        last_ba["its"] = np.array([sum(r["stats"]["n_its"]) for r in res], np.int32)
        last_ba["pose"] = shard.rt_to_quat_t(np.concatenate([r["kf_pose"] for r in res])).reshape(B, shard.SLOT_BA_KF, shard.SLOT_BA_DOUBLES)
        return int(last_ba["its"].sum())

    def step(k):
        cur, prv = k % NBUF, (k - 1) % NBUF
        if two_streams:
            tstream.wait_event(ev_m[cur])     # buffer `cur` was the "previous frame" of match(k - 2): let it finish first
        ext.extract_batch_dev(d_frames[k & 1], d_kps[cur], d_desc[cur], d_n[cur], stream=stream)
        if two_streams:
            ev_x[cur].record(tstream)
            mstream.wait_event(ev_x[cur])
        api.ORBmatcher.best2_batch_dev(d_desc[cur], d_n[cur], d_desc[prv], d_n[prv], d_idx, d_best, d_second,
                                       cap, device=dev.index, stream=mstream.cuda_stream)
        if two_streams:
            ev_m[prv].record(mstream)
        if pipeline:
            ba_jobs.append(ba.submit(ba_probs))   # this step's windows; never waited for here
        if (world > 1 or args.force_dist) and not pipeline:  # result summary to every rank (the only cross-GPU traffic of the path)
            if cdev.type == "cpu":
                gather.local[:, 0] = d_n[cur].cpu()
            else:
                gather.local[:, 0] = d_n[cur]
            gather.step()
        if pipeline:
            # fixed-capacity result slots: header, first 2000 keypoints + descriptors, and the 50 poses of the BA batch that was
            # submitted --ba-workers steps ago (the oldest one in flight: usually long done)
            if len(ba_jobs) > ba.n:
                ba_its[0] += ba_take(ba.result(ba_jobs.pop(0)))
            j = k % NPIN
            pin_ev[j].synchronize()   # (returns at once unless the stream is four steps behind)
            pin_its[j].numpy()[:] = last_ba["its"]
            pin_pose[j].numpy()[:] = last_ba["pose"]
            on_host = cdev.type == "cpu"   # gloo rehearsal: the slots live on the host
            hdr, kp, ds, bp = slots.header(), slots.keypoints(), slots.descriptors(), slots.ba_poses()
            n_cur = d_n[cur]
            acc = ((d_best <= 50) & (d_best.float() < 0.9 * d_second.float()) & (d_col < n_cur[:, None])).sum(1, dtype=torch.int32)
            if on_host:
                hdr[:, 0].copy_(torch.clamp(n_cur, max=shard.SLOT_KP_CAP).cpu()); hdr[:, 1].copy_(acc.cpu()); hdr[:, 2].copy_(pin_its[j]); hdr[:, 3].copy_(d_sid.cpu())
                kp.copy_(d_kps[cur][:, :shard.SLOT_KP_CAP].cpu()); ds.copy_(d_desc[cur][:, :shard.SLOT_KP_CAP].cpu()); bp.copy_(pin_pose[j])
            else:   # device to device, plus two small pinned-to-device copies on the step's stream
                hdr[:, 0].copy_(torch.clamp(n_cur, max=shard.SLOT_KP_CAP)); hdr[:, 1].copy_(acc); hdr[:, 2].copy_(pin_its[j], non_blocking=True); hdr[:, 3].copy_(d_sid)
                kp.copy_(d_kps[cur][:, :shard.SLOT_KP_CAP]); ds.copy_(d_desc[cur][:, :shard.SLOT_KP_CAP]); bp.copy_(pin_pose[j], non_blocking=True)
                pin_ev[j].record(tstream)
            slots.step()

    def ba_drain():
        """Waits for every BA batch still in flight (the timed region ends with its work done); returns their LM iterations."""
        its = 0
        while pipeline and ba_jobs:
            its += ba_take(ba.result(ba_jobs.pop(0)))
        return its

    for k in range(args.warmup):
        step(k)
    ba_drain()
    torch.cuda.synchronize(dev)
    ba_its[0] = 0
    ext.profile(5)   # timed regions: events around the dominant kernel (FAST) only, on every 4th step
    m0 = torch.cuda.Event(enable_timing=True)
    m1 = torch.cuda.Event(enable_timing=True)
    match_ms = 0.0
    reps = max(1, args.reps if not pipeline else min(args.reps, 3))
    regions, ba_regions = [], []
    last_slots = None
    k_next = args.warmup
    for rep in range(reps):   # every region: EXACTLY --steps steps between barrier + synchronize pairs
        ba_its[0] = 0
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(k_next, k_next + args.steps):
            step(k)
        k_next += args.steps
        if (world > 1 or args.force_dist) and not pipeline:
            gather.flush()   # the last step's summary is part of the timed work
        if pipeline:
            last_slots = slots.flush()
            ba_its[0] += ba_drain()   # the region's BA batches are part of its work: all of them finish inside it
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        regions.append(shard.max_over_ranks(time.perf_counter() - t0, cdev, world))
        ba_regions.append(ba_its[0])
    mid = sorted(range(reps), key=lambda i: regions[i])[reps // 2]   # the median region (same index on every rank: the times are the rank maxima)
    elapsed = regions[mid]
    ba_its_timed = ba_regions[mid]
    stages = ext.profile(1)
    # per-stage breakdown: a separate, untimed pass with events around every stage
    for k in range(k_next, k_next + (1 if pipeline else 5)):
        step(k)
    if (world > 1 or args.force_dist) and not pipeline:
        gather.flush()
    if pipeline:
        slots.flush()
        ba_drain()
    torch.cuda.synchronize(dev)
    all_stages = ext.profile(0)

    # matcher kernel time, measured separately on the same stream (it is one launch per step)
    m0.record(torch.cuda.current_stream(dev))
    for k in range(5):
        api.ORBmatcher.best2_batch_dev(d_desc[0], d_n[0], d_desc[1], d_n[1], d_idx, d_best, d_second, cap,
                                       device=dev.index, stream=stream)
    m1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    match_ms = m0.elapsed_time(m1) / 5

    # sanity of the timed work: every frame produced its keypoints and matches
    n_last = d_n[(k_next - 1) % NBUF].cpu().numpy()
    assert (n_last >= NFEAT).all(), "extractor returned too few keypoints: %s" % n_last[:8]
    if pipeline:   # every stream's slot arrived on this rank, with its keypoints, descriptors and BA poses
        hdr = slots.header(last_slots).cpu().numpy()
        assert sorted(hdr[:, 3].tolist()) == list(range(args.streams)), hdr[:, 3]
        assert (hdr[:, 0] == shard.SLOT_KP_CAP).all() and (hdr[:, 2] > 0).all(), hdr[:, :3]
        q = slots.ba_poses(last_slots)[:, :, :4].cpu().numpy()
        assert np.allclose((q * q).sum(2), 1.0, atol=1e-9), "BA slot poses are not unit quaternions"

    if rank == 0:
        fast_ms, fast_calls = stages["fast"]
        fast_avg_ms = fast_ms / max(fast_calls, 1)
        achieved = FAST_BYTES_PER_FRAME * B / (fast_avg_ms * 1e-3) / 1e9 if fast_avg_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic(B) if args.config == "vga" else (None, "only collected for the vga configuration")
        out = {
            "metric": ("frames/sec ORB extract+match @%dx%dx8lvl" % (W, H)) + (" + per-stream local BA, %d streams" % args.streams if pipeline else ""),
            "value": round(world * B * args.steps / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "repetitions": reps,
            "repetitions_ms_per_step": [round(1e3 * t / args.steps, 4) for t in regions],
            "timing": "median of %d repetitions of the %d-step region, each bracketed by barrier + synchronize (max over ranks)" % (reps, args.steps),
            "higher_is_better": True,
            "scaling": "strong" if pipeline else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if args.scene == "rich" else "synthetic (scene=%s)" % args.scene,
            "config": {"workload": "BASELINE %s: %dx%d 8-level ORB extract (%d features, FAST 20/7) + "
                                   "brute-force 256-bit Hamming best/second match vs the stream's previous frame; "
                                   "%d independent streams per GPU per step" % (cfg_name, W, H, NFEAT, B) +
                                   ("; every stream also solves one local-BA window (50 KF, 2000 points, window-8) per step on the BA "
                                    "handle's own HIP stream (%d batches in flight per rank: a slot's poses come from the batch submitted that many steps "
                                    "earlier); every stream's result slot (count + 2000 keypoints + 2000 descriptors + "
                                    "50 poses, %d B) is all_gathered to every rank, one step late" % (args.ba_workers, shard.SLOT_BYTES) if pipeline else ""),
                       "frames_per_step_per_gpu": B, "distinct_frame_pairs_per_gpu": uniq,
                       "collective": (args.backend if (world > 1 or args.force_dist) else None),
                       "ba_batches_in_flight": args.ba_workers if pipeline else None,
                       "parallelism": ("stream s on rank s %% %d, results gathered over RCCL" % world) if pipeline else
                                      "streams sharded over %d GPU(s), no data-path collective" % world},
            "roofline": {"kernel": "fast_cells_kernel", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": round(fast_avg_ms, 5), "launches": fast_calls,
                         "timing": "HIP events on the launch stream around every 4th launch of the timed region (an event pair drains the pipeline for ~20 us)",
                         "algorithmic_bytes_per_launch": FAST_BYTES_PER_FRAME * B,
                         "valu": measured_valu(B, fast_avg_ms, FAST_BYTES_PER_FRAME) if args.config == "vga" else None},
            "stage_ms_per_step": {k: round(v[0] / max(v[1], 1), 4) for k, v in all_stages.items()},
            "match_ms_per_step": round(match_ms, 4),
        }
        if pipeline:
            out["secondary"] = {"metric": "local-BA LM iterations/sec inside the pipeline (one window-8 window per stream and step)",
                                "unit": "iters/s", "value": round(world * ba_its_timed / elapsed, 1), "note": "this rank's count x ranks"}
        if not args.no_cpu and world == 1:   # CPU comparators: rank 0 at N = 1 only
            out["cpu"] = cpu_info()
            out["cpu_baseline"] = cpu_baseline(fa, fb, NFEAT)
            out["speedup_vs_cpu_1core"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(fa, fb, NFEAT)
        if world == 1 and args.config == "vga" and not args.no_extras:
            # the driver's record also carries: the geometry the >= 70 % bar is stated on, and the host-pointer latencies
            try:
                out["roofline_720p"] = roofline_720p(dev)
            except Exception as e:
                out["roofline_720p"] = {"error": repr(e)}
            try:
                out["latency"] = host_api_latency(fa, fb)
            except Exception as e:
                out["latency"] = {"error": repr(e)}
        if not args.no_ba and world == 1 and not pipeline:
            out["secondary"] = ba_secondary(dev.index, args.steps, with_cpu=not args.no_cpu)
        print(json.dumps(out), flush=True)
    if pipeline:
        ba.close()
    if world > 1:
        dist.barrier()
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
