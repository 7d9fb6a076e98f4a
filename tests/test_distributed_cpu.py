"""World-size-2 gloo test of the N>1 path on CPU: env sharding arithmetic + the ONE all-gather of the packed compact records + the record
layout the consumer unpacks.  Each rank steps its shard with the CPU oracle (stand-in for the per-rank HIP stepper, which needs a GPU --
tests/test_gpu_distributed.py runs the same flow with the HIP stepper on both ranks); rank 0 checks the gathered batch against a
single-process run."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _step(e, key):
    """One auto-resetting step of an oracle env -> (reward, done) the way parallel.py:59-64 derives them from the returned state: the positive
    gold delta and the terminal flag."""
    g0 = int(e.status_arr()[1])
    e.step_autoreset(key)
    return float(max(0, int(e.status_arr()[1]) - g0)), e.flags()["is_terminal"]


def _worker(rank, world, port, n_total, steps, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "rogue-gym_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import OracleEnv
    from rogue_gym.envs.sharding import all_gather_packed, record_layout, shard_range, unpack_records, unpack_step

    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))["configs"]["mini"]
    first, last = shard_range(n_total, rank, world)
    envs = [OracleEnv(cfg, seed=i, max_steps=20) for i in range(first, last)]
    acts = np.frombuffer(b".hjklnbuy>s", np.uint8)
    rng = np.random.RandomState(0)
    last = [(0.0, False)] * len(envs)
    for _ in range(steps):
        keys = acts[rng.randint(0, 11, n_total)]  # same global action tensor on every rank
        for j, e in enumerate(envs):
            last[j] = _step(e, int(keys[first + j]))
    # the record rg_pack_compact writes per env: screen bytes, status i32[10], reward f32, flags u32, history bytes
    o_scr, o_st, o_hist, rec = record_layout(16, 32, with_hist=True)
    packed = np.zeros((len(envs), rec), np.uint8)
    for j, e in enumerate(envs):
        packed[j, o_scr:o_st] = e.screen().reshape(-1)
        packed[j, o_st:o_st + 40] = e.status_arr().astype(np.int32).view(np.uint8)
        packed[j, o_st + 40:o_st + 44] = np.array([last[j][0]], np.float32).view(np.uint8)
        packed[j, o_st + 44:o_st + 48] = np.array([1 if last[j][1] else 0], np.uint32).view(np.uint8)
        packed[j, o_hist:] = e.hist().reshape(-1)
    gathered = all_gather_packed(torch.from_numpy(packed))  # ONE collective
    scr, st, hist = unpack_records(gathered, 16, 32, with_hist=True)
    rew, done, _flags = unpack_step(gathered, 16, 32, with_hist=True)
    if rank == 0:
        q.put((scr.numpy(), st.numpy(), hist.numpy(), rew.numpy(), done.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_partition():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "rogue-gym_amd"))
    from rogue_gym.envs.sharding import shard_range
    for n, w in [(65536, 8), (262144, 8), (10000, 8), (7, 2), (64, 1)]:
        r = [shard_range(n, i, w) for i in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
    assert shard_range(10000, 3, 8) == (3750, 5000)  # BASELINE config 5: 1 250 envs per GPU


@pytest.mark.timeout(120)
def test_world2_gloo_gather_matches_single_process():
    import sys
    sys.path.insert(0, ROOT)
    from oracle.pyoracle import OracleEnv
    n_total, steps, world = 24, 30, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    scr, st, hist, rew, done = q.get(timeout=100)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))["configs"]["mini"]
    envs = [OracleEnv(cfg, seed=i, max_steps=20) for i in range(n_total)]
    acts = np.frombuffer(b".hjklnbuy>s", np.uint8)
    rng = np.random.RandomState(0)
    last = [(0.0, False)] * n_total
    for _ in range(steps):
        keys = acts[rng.randint(0, 11, n_total)]
        for i, e in enumerate(envs):
            last[i] = _step(e, int(keys[i]))
    for i, e in enumerate(envs):
        assert np.array_equal(scr[i], e.screen())
        assert np.array_equal(st[i], e.status_arr().astype(np.int32))
        assert np.array_equal(hist[i], e.hist())
        # the step half of the ONE collective: reward and done of every env of the job (thread_impls.rs:61-81, parallel.py:59-64)
        assert rew[i] == np.float32(last[i][0]) and bool(done[i]) == bool(last[i][1]), i


@pytest.mark.timeout(60)
def test_spawn_ranks_kills_the_survivors_when_one_rank_dies(tmp_path, monkeypatch):
    """bench.py's own launcher (`python bench.py --gpus N` without torchrun): when one rank exits with an error the others -- blocked in a barrier that
    will never complete -- are killed and the worst exit code is returned, instead of the driver's command hanging."""
    import sys
    import time
    sys.path.insert(0, ROOT)
    import bench

    script = tmp_path / "fake_rank.py"
    script.write_text("import os, sys, time\n"
                      "open(os.path.join(os.path.dirname(__file__), 'pid%s' % os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                      "assert os.environ['WORLD_SIZE'] == '4' and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                      "if os.environ['RANK'] == '2':\n    time.sleep(0.5); sys.exit(7)\n"
                      "time.sleep(600)\n")
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setattr(sys, "argv", [str(script)])
    t0 = time.time()
    rc = bench.spawn_ranks(4)
    assert rc == 7 and time.time() - t0 < 30
    time.sleep(0.3)
    for r in range(4):
        pid = int((tmp_path / ("pid%d" % r)).read_text())
        try:
            os.kill(pid, 0)
            alive = open("/proc/%d/stat" % pid).read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, "rank %d (pid %d) survived" % (r, pid)
