"""The N>1 path on CPU: world_size 2, gloo.  Covers what bench.py does across ranks — stream
sharding, the per-step summary all_gather, barrier + max-over-ranks timing — without a GPU."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from weiner_slamit_v2_amd import shard

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert shard.env_rank() == (rank, rank, world)
        per_rank = 4
        mine = shard.weak_streams(per_rank, world, rank)
        g = shard.SummaryGather(per_rank, 2, torch.device("cpu"), world)
        def check(allv, step):
            assert allv.shape == (world * per_rank, 2)
            assert allv[:, 0].tolist() == [1000 + s for s in range(world * per_rank)]
            assert allv[:, 1].tolist() == [s * 10 + step for s in range(world * per_rank)]

        for step in range(4):
            for i, s in enumerate(mine):  # pretend: stream s produced 1000+s keypoints, s matches at this step
                g.local[i, 0] = 1000 + s
                g.local[i, 1] = s * 10 + step
            prev = g.step()               # pipelined by one step: returns the summary of step - 1
            if step == 0:
                assert prev is None
            else:
                check(prev, step - 1)
        check(g.flush(), 3)
        assert g.flush() is None
        dist.barrier()
        t = shard.max_over_ranks(1.0 + rank, torch.device("cpu"), world)
        assert t == float(world)
        strong = shard.stream_assignment(8, world, rank)
        assert strong == list(range(rank, 8, world))
        # fixed-capacity result slots of the 8-stream pipeline: layout + one-step-late gather
        assert shard.SLOT_BYTES % 256 == 0 and shard.SLOT_BYTES >= 16 + 2000 * (28 + 32) + 50 * 7 * 8
        sg = shard.SlotGather(len(strong), torch.device("cpu"), world)
        def fill(step):
            h = sg.header()
            for i, sid in enumerate(strong):
                h[i, 0], h[i, 1], h[i, 2], h[i, 3] = 1900 + sid, 100 * step + sid, step, sid
                sg.keypoints()[i, :, 0] = float(sid) + 0.5
                sg.keypoints()[i, :, 5].view(torch.int32)[:] = step
                sg.descriptors()[i, :, :] = (7 * sid + step) % 256
                sg.ba_poses()[i, :, :] = sid + step / 16.0
        def check_slots(allb, step):
            assert allb.shape == (8, shard.SLOT_BYTES)
            h = sg.header(allb)
            for row in range(8):
                sid = int(h[row, 3])
                assert sid == (row // len(strong)) + world * (row % len(strong))   # rank-major rows, stream s on rank s % world
                assert h[row].tolist() == [1900 + sid, 100 * step + sid, step, sid]
                assert float(sg.keypoints(allb)[row, 1999, 0]) == sid + 0.5
                assert int(sg.keypoints(allb)[row, 0, 5].view(torch.int32)) == step
                assert int(sg.descriptors(allb)[row, 1999, 31]) == (7 * sid + step) % 256
                assert float(sg.ba_poses(allb)[row, 49, 6]) == sid + step / 16.0
        for step in range(3):
            fill(step)
            prev = sg.step()
            if step:
                check_slots(prev, step - 1)
            else:
                assert prev is None
        check_slots(sg.flush(), 2)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_assignments_cover_every_stream_once():
    sys.path.insert(0, ROOT)
    from weiner_slamit_v2_amd import shard

    for world in (1, 2, 4, 8):
        seen = sorted(s for r in range(world) for s in shard.stream_assignment(8, world, r))
        assert seen == list(range(8))
        seen = sorted(s for r in range(world) for s in shard.weak_streams(64, world, r))
        assert seen == list(range(64 * world))


def test_rt_to_quat_t_round_trip_including_half_turns():
    """shard.rt_to_quat_t (the BA slot's pose format): quaternion -> R must give the input back for every rotation, the
    180-degree ones (w = 0, where the off-diagonal differences vanish) included; numpy and torch inputs agree."""
    import numpy as np
    import torch

    from weiner_slamit_v2_amd import shard

    def q2R(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    rs = np.random.RandomState(0)
    qs = rs.randn(64, 4)
    qs[:16, 3] = 0                                         # half turns about random axes
    qs[16] = [1 / np.sqrt(2), -1 / np.sqrt(2), 0, 0]       # the axis (1, -1, 0) / sqrt 2 of the review
    qs /= np.linalg.norm(qs, axis=1)[:, None]
    R = np.stack([q2R(q) for q in qs])
    rt = np.concatenate([R.reshape(-1, 9), rs.randn(64, 3)], 1)
    out = shard.rt_to_quat_t(rt)
    assert np.allclose((out[:, :4] ** 2).sum(1), 1.0, atol=1e-12) and (out[:, 3] >= 0).all()
    assert max(np.abs(q2R(o[:4]) - R[i]).max() for i, o in enumerate(out)) < 1e-12
    assert np.array_equal(out[:, 4:], rt[:, 9:])
    assert np.allclose(shard.rt_to_quat_t(torch.from_numpy(rt)).numpy(), out, atol=0, rtol=0)
