"""Multi-GPU plumbing of the hot path: one process per GPU, independent camera streams / BA windows
sharded over ranks, no collective on the data path.  The only cross-rank traffic
is a per-frame result summary gathered to every rank with one small all_gather per step
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests).
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def stream_assignment(n_streams, world, rank):
    """Stream s runs on rank s % world: returns this rank's stream ids, ascending."""
    return list(range(rank, n_streams, world))


def weak_streams(per_rank, world, rank):
    """Weak scaling: every rank owns `per_rank` streams; global ids are rank-major."""
    return list(range(rank * per_rank, (rank + 1) * per_rank))


class SummaryGather:
    """Gathers a small int32 summary (rows = this rank's frames) from all ranks each step.

    Pipelined by one step: step() launches the all_gather of what the caller just wrote into `local` asynchronously
    (on the backend's own stream, ordered after the caller's stream) and returns the result of the PREVIOUS step's
    gather, so the collective's latency overlaps the next step's kernels instead of sitting on the compute stream.
    `local` / the gathered tensors are double-buffered; flush() returns the newest result."""

    def __init__(self, rows, cols, device, world, force_collective=False):
        self.world = world
        # force_collective: run the all_gather even on one rank (an initialised process group of size 1), so that the RCCL path of the
        # default bench is exercised by a one-GPU test before a node sees it
        self.collective = world > 1 or force_collective
        self._loc = [torch.zeros((rows, cols), dtype=torch.int32, device=device) for _ in range(2)]
        self._all = [torch.zeros((world * rows, cols), dtype=torch.int32, device=device) for _ in range(2)] if self.collective else self._loc
        self._k = 0
        self._pending = None
        self.local = self._loc[0]

    def step(self):
        """Publishes `local`; returns the fleet summary of the previous step (None on the first call)."""
        prev = self.flush() if (self._pending is not None or not self.collective and self._k > 0) else None
        cur = self._k & 1
        if self.collective:
            self._pending = (dist.all_gather_into_tensor(self._all[cur], self._loc[cur], async_op=True), self._all[cur])
        else:
            self._pending = (None, self._loc[cur])
        self._k += 1
        self.local = self._loc[self._k & 1]
        return prev

    def flush(self):
        """Waits for the gather in flight (stream-ordered on GPU backends) and returns its result."""
        if self._pending is None:
            return None
        work, out = self._pending
        if work is not None:
            work.wait()
        self._pending = None
        return out


# ---- fixed-capacity result slots: what a rank publishes per stream and step ----------------------
SLOT_KP_CAP = 2000                      # keypoints kept per frame (the extractor may return up to nfeatures + 3 per level)
SLOT_KP_BYTES = 28                      # slamit_kp == cv::KeyPoint
SLOT_DESC_BYTES = 32
SLOT_BA_KF = 50                         # keyframes of a local-BA window
SLOT_BA_DOUBLES = 7                     # pose as unit quaternion (x, y, z, w) + translation
SLOT_OFF_COUNT = 0                      # int32 keypoints, int32 matches, int32 BA iterations, int32 stream id
SLOT_OFF_KP = 16
SLOT_OFF_DESC = SLOT_OFF_KP + SLOT_KP_CAP * SLOT_KP_BYTES
SLOT_OFF_BA = SLOT_OFF_DESC + SLOT_KP_CAP * SLOT_DESC_BYTES
SLOT_BYTES = (SLOT_OFF_BA + SLOT_BA_KF * SLOT_BA_DOUBLES * 8 + 255) // 256 * 256


class SlotGather:
    """Every rank's per-stream result slots to every rank, one `all_gather_into_tensor` per step, issued asynchronously and
    consumed one step late (like SummaryGather).  A slot is SLOT_BYTES of uint8:
        [0:16)   int32 x 4: keypoint count (clamped to SLOT_KP_CAP), accepted matches, BA iterations, global stream id
        [16: )   SLOT_KP_CAP keypoints of 28 bytes (x, y, size, angle, response: float32; octave, class_id: int32)
        then     SLOT_KP_CAP descriptors of 32 bytes
        then     SLOT_BA_KF x 7 float64: the window's optimised keyframe poses (quaternion x y z w, translation)
    No reduction is needed for correctness: streams are independent, the gather only makes every result visible everywhere."""

    def __init__(self, rows, device, world, force_collective=False):
        self.world, self.rows = world, rows
        # force_collective: run the all_gather even on one rank (an initialised process group of size 1): the RCCL path -- communicator,
        # device-tensor all_gather_into_tensor, async work handle -- is then exercised by a one-GPU test before a node sees it
        self.collective = world > 1 or force_collective
        self._loc = [torch.zeros((rows, SLOT_BYTES), dtype=torch.uint8, device=device) for _ in range(2)]
        self._all = [torch.zeros((world * rows, SLOT_BYTES), dtype=torch.uint8, device=device) for _ in range(2)] if self.collective else self._loc
        self._k = 0
        self._pending = None

    @property
    def local(self):
        return self._loc[self._k & 1]

    def header(self, buf=None):
        """int32 view [rows, 4] of the slots' headers (of `buf`, default the local slots being filled)."""
        b = self.local if buf is None else buf
        return b[:, SLOT_OFF_COUNT:SLOT_OFF_KP].view(torch.int32)

    def keypoints(self, buf=None):
        b = self.local if buf is None else buf
        return b[:, SLOT_OFF_KP:SLOT_OFF_DESC].view(torch.float32).view(b.shape[0], SLOT_KP_CAP, 7)

    def descriptors(self, buf=None):
        b = self.local if buf is None else buf
        return b[:, SLOT_OFF_DESC:SLOT_OFF_BA].view(b.shape[0], SLOT_KP_CAP, SLOT_DESC_BYTES)

    def ba_poses(self, buf=None):
        b = self.local if buf is None else buf
        return b[:, SLOT_OFF_BA:SLOT_OFF_BA + SLOT_BA_KF * SLOT_BA_DOUBLES * 8].view(torch.float64).view(b.shape[0], SLOT_BA_KF, SLOT_BA_DOUBLES)

    def step(self):
        """Publishes the local slots; returns every rank's slots of the PREVIOUS step (None on the first call)."""
        prev = self.flush()
        cur = self._k & 1
        if self.collective:
            self._pending = (dist.all_gather_into_tensor(self._all[cur], self._loc[cur], async_op=True), self._all[cur])
        else:
            self._pending = (None, self._loc[cur])
        self._k += 1
        return prev

    def flush(self):
        if self._pending is None:
            return None
        work, out = self._pending
        if work is not None:
            work.wait()
        self._pending = None
        return out


class BaWorkers:
    """The pipeline's local-BA side: `nworkers` host threads, each with its own BA handle (own HIP stream and pinned block), take
    the rank's per-step window batches in turn.  A solve of eight windows is latency bound (~4 ms of small dependent launches
    that leave the chip mostly idle), so several batches IN FLIGHT beside the extractor's launches multiply the throughput;
    submit() never blocks on the solve it queues, result() of a job blocks until that job is done.  A step's slot therefore
    carries the poses of the batch submitted `nworkers` steps earlier (stated in the bench line)."""

    def __init__(self, make_optimizer, nworkers):
        import queue
        import threading

        self.n = max(1, int(nworkers))
        self._q = [queue.Queue() for _ in range(self.n)]
        self._done = {}
        self._cv = threading.Condition()
        self._submitted = 0
        self._err = None

        def run(i):
            try:
                opt = make_optimizer()
                while True:
                    job = self._q[i].get()
                    if job is None:
                        break
                    jid, probs = job
                    res = opt.LocalBundleAdjustmentBatch(probs)   # ctypes releases the GIL inside the solve
                    with self._cv:
                        self._done[jid] = res
                        self._cv.notify_all()
                opt.close()
            except Exception as e:   # surfaced by result()
                with self._cv:
                    self._err = e
                    self._cv.notify_all()

        self._threads = [threading.Thread(target=run, args=(i,), daemon=True) for i in range(self.n)]
        for t in self._threads:
            t.start()

    def submit(self, probs):
        jid = self._submitted
        self._q[jid % self.n].put((jid, probs))
        self._submitted += 1
        return jid

    def ready(self, jid):
        with self._cv:
            return jid in self._done

    def result(self, jid, keep=False):
        with self._cv:
            while jid not in self._done and self._err is None:
                self._cv.wait()
            if self._err is not None:
                raise self._err
            return self._done[jid] if keep else self._done.pop(jid)

    def close(self):
        for q in self._q:
            q.put(None)
        for t in self._threads:
            t.join(30)


def rt_to_quat_t(rt):
    """n x 12 (R row-major | t) -> n x 7 (quaternion x y z w | t); the BA slot's pose format.  numpy array or torch tensor in,
    the same kind out (a tensor stays on its device: the pipeline packs its slots without a host round trip).  Shepperd's
    method: the largest of (trace, R00, R11, R22) picks the branch, so rotations by 180 degrees (w = 0) come out right --
    copysign of a zero difference would not."""
    import numpy as np

    is_t = isinstance(rt, torch.Tensor)
    if not is_t:   # host arrays: plain numpy (a dozen small array operations; the same through torch costs 0.3 ms of dispatch per call)
        X = np.asarray(rt, np.float64).reshape(-1, 12)
        r00, r11, r22 = X[:, 0], X[:, 4], X[:, 8]
        tr = r00 + r11 + r22
        br = np.argmax(np.stack([tr, r00, r11, r22], 1), 1)
        s = np.sqrt(np.maximum(np.stack([1.0 + tr, 1.0 + r00 - r11 - r22, 1.0 - r00 + r11 - r22, 1.0 - r00 - r11 + r22], 1), 0.0)) * 2   # (the tensor path's expressions, bit for bit)
        s4 = s / 4
        s = s + 1e-300
        d21, d02, d10 = X[:, 7] - X[:, 5], X[:, 2] - X[:, 6], X[:, 3] - X[:, 1]
        a01, a02, a12 = X[:, 1] + X[:, 3], X[:, 2] + X[:, 6], X[:, 5] + X[:, 7]
        qs = np.stack([np.stack([d21 / s[:, 0], d02 / s[:, 0], d10 / s[:, 0], s4[:, 0]], 1),
                       np.stack([s4[:, 1], a01 / s[:, 1], a02 / s[:, 1], d21 / s[:, 1]], 1),
                       np.stack([a01 / s[:, 2], s4[:, 2], a12 / s[:, 2], d02 / s[:, 2]], 1),
                       np.stack([a02 / s[:, 3], a12 / s[:, 3], s4[:, 3], d10 / s[:, 3]], 1)], 1)
        q = qs[np.arange(len(X)), br]
        q = np.where(q[:, 3:4] < 0, -q, q)
        return np.concatenate([q, X[:, 9:12]], 1)
    X = rt.to(torch.float64).reshape(-1, 12)
    R = X[:, :9].reshape(-1, 3, 3)
    r00, r11, r22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    tr = r00 + r11 + r22
    cand = torch.stack([tr, r00, r11, r22], 1)
    br = cand.argmax(1)
    q = torch.zeros((len(X), 4), dtype=torch.float64, device=X.device)
    # branch 0: w largest
    s0 = torch.sqrt(torch.clamp(1.0 + tr, min=0.0)) * 2
    s1 = torch.sqrt(torch.clamp(1.0 + r00 - r11 - r22, min=0.0)) * 2
    s2 = torch.sqrt(torch.clamp(1.0 - r00 + r11 - r22, min=0.0)) * 2
    s3 = torch.sqrt(torch.clamp(1.0 - r00 - r11 + r22, min=0.0)) * 2
    eps = 1e-300
    qs = [
        torch.stack([(R[:, 2, 1] - R[:, 1, 2]) / (s0 + eps), (R[:, 0, 2] - R[:, 2, 0]) / (s0 + eps), (R[:, 1, 0] - R[:, 0, 1]) / (s0 + eps), s0 / 4], 1),
        torch.stack([s1 / 4, (R[:, 0, 1] + R[:, 1, 0]) / (s1 + eps), (R[:, 0, 2] + R[:, 2, 0]) / (s1 + eps), (R[:, 2, 1] - R[:, 1, 2]) / (s1 + eps)], 1),
        torch.stack([(R[:, 0, 1] + R[:, 1, 0]) / (s2 + eps), s2 / 4, (R[:, 1, 2] + R[:, 2, 1]) / (s2 + eps), (R[:, 0, 2] - R[:, 2, 0]) / (s2 + eps)], 1),
        torch.stack([(R[:, 0, 2] + R[:, 2, 0]) / (s3 + eps), (R[:, 1, 2] + R[:, 2, 1]) / (s3 + eps), s3 / 4, (R[:, 1, 0] - R[:, 0, 1]) / (s3 + eps)], 1),
    ]
    for i in range(4):
        q = torch.where((br == i)[:, None], qs[i], q)
    q = torch.where((q[:, 3] < 0)[:, None], -q, q)   # w >= 0, like SE3Quat::normalizeRotation (se3quat.h:276-281)
    return torch.cat([q, X[:, 9:12]], 1)


def max_over_ranks(value, device, world):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
