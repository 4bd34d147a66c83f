"""bench.py contract checks on the GPU box: the JSON line's schema at N = 1 and the multi-rank flow rehearsed with two
ranks sharing the GPU (gloo carries the summary; the real run uses nccl = RCCL with one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    return json.loads(lines[0])


def test_bench_line_schema_single_gpu():
    out = subprocess.check_output([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--no-cpu", "--no-ba"], cwd=ROOT,
                                  stderr=subprocess.STDOUT, timeout=300)
    d = _line(out)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "u8" and d["scaling"] == "weak" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["launches"] >= 1
    frames = d["config"]["frames_per_step_per_gpu"]
    assert frames == 256 and abs(d["value"] - frames * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 0.01


def test_bench_two_rank_rehearsal():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29713", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--no-cpu", "--no-ba"]
    out = subprocess.check_output(cmd, cwd=ROOT, stderr=subprocess.STDOUT, timeout=600, env=env)
    d = _line(out)
    assert d["n_gpus"] == 2 and d["steps"] == 3
    frames = d["config"]["frames_per_step_per_gpu"]
    assert abs(d["value"] - 2 * frames * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 0.01   # whole-job aggregate over both ranks


def test_bench_pipeline_config_single_gpu_and_two_rank_rehearsal():
    """BASELINE configs[4] as bench.py runs it: 8 x 720p streams dealt over the ranks (stream s on rank s % N), every rank
    also solving its streams' local-BA windows beside the extractor, every stream's fixed-capacity result slot gathered
    to every rank.  bench.py itself asserts that all 8 slots arrived with 2000 keypoints and unit-quaternion BA poses."""
    out = subprocess.check_output([sys.executable, "bench.py", "--config", "pipeline", "--steps", "2", "--warmup", "1", "--no-cpu"], cwd=ROOT,
                                  stderr=subprocess.STDOUT, timeout=600)
    d = _line(out)
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["config"]["frames_per_step_per_gpu"] == 8
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.01 and d["secondary"]["value"] > 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29717", "bench.py", "--gpus", "2", "--config", "pipeline", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu"]
    out = subprocess.check_output(cmd, cwd=ROOT, stderr=subprocess.STDOUT, timeout=900, env=env)
    d = _line(out)
    assert d["n_gpus"] == 2 and d["config"]["frames_per_step_per_gpu"] == 4
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.01   # 8 streams in all, whatever the rank count


def test_bench_pipeline_forced_collective_on_one_gpu():
    """--force-dist: an initialised nccl (= RCCL) process group of size one, so the pipeline's all_gather_into_tensor of the
    device-resident slots, its async work handle and the max-over-ranks all_reduce run through RCCL on this one GPU before an
    8-GPU node sees them.  bench.py's own asserts (8 slots, 2000 keypoints, unit quaternions, BA iterations > 0) still hold."""
    out = subprocess.check_output([sys.executable, "bench.py", "--config", "pipeline", "--force-dist", "--steps", "6", "--warmup", "2",
                                   "--reps", "1", "--no-cpu"], cwd=ROOT, stderr=subprocess.STDOUT, timeout=600)
    d = _line(out)
    assert d["n_gpus"] == 1 and d["config"]["frames_per_step_per_gpu"] == 8 and d["config"].get("collective") == "nccl"
    assert d["secondary"]["value"] > 0


def test_bench_default_config_forced_collective_on_one_gpu():
    """The driver's multi-GPU run of the DEFAULT config differs from N = 1 only by the process group, the per-step
    all_gather_into_tensor of the [frames, 2] summary and the max-over-ranks all_reduce: --force-dist runs exactly those through
    nccl (= RCCL) at world size one, so the code a node will execute has run on a GPU before."""
    out = subprocess.check_output([sys.executable, "bench.py", "--force-dist", "--steps", "4", "--warmup", "1", "--reps", "2", "--no-cpu", "--no-ba",
                                   "--no-extras"], cwd=ROOT, stderr=subprocess.STDOUT, timeout=600)
    d = _line(out)
    assert d["n_gpus"] == 1 and d["config"].get("collective") == "nccl" and d["config"]["frames_per_step_per_gpu"] == 256
    assert abs(d["value"] - 256 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 0.01
