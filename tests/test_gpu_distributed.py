"""World-size-2 run of the HIP stepper itself (two processes on GPU 0, gloo rendezvous -- the 1-GPU stand-in for two RCCL ranks): every
rank steps ITS shard with librogue_gym_hip.so, the ONE all-gather of packed compact records assembles the whole-job batch, the HIP expand
kernels turn it into f32 images, and the result equals a single-process HIP run over all envs (BASELINE configs 4 and 5: symbol image
of the nohide dungeon; status + gray of per-env seeds).  Also drives bench.py's N > 1 control flow with ROGUE_GYM_BENCH_ONE_DEVICE=1."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, kind, n_total, steps, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "rogue-gym_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rogue_gym.envs import DungeonType, HipVecRogueEnv, ImageSetting, StatusFlag
    from rogue_gym.envs.sharding import shard_range

    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))["configs"][name]
    first, last = shard_range(n_total, rank, world)
    st = ImageSetting(DungeonType.SYMBOL if kind == "symbol" else DungeonType.GRAY, StatusFlag.FULL, kind != "symbol")
    env = HipVecRogueEnv([dict(cfg, seed=i) for i in range(first, last)], max_steps=40, image_setting=st, device=0)
    g = torch.Generator(device="cpu").manual_seed(5)
    for _ in range(steps):
        a = torch.randint(0, 11, (n_total,), generator=g)  # the same global action tensor on every rank
        env.step(a[first:last].to(env.device))
    full = env.all_gather_obs(compact=True)     # ONE collective + HIP expand
    raw = env.all_gather_obs(compact=False)     # the f32 gather of the literal config text
    scr, status, hist = env.all_gather_compact(with_hist=True)
    obs2, rew, done, flags = env.all_gather_step()  # the whole step of the whole job from the same ONE collective
    torch.cuda.synchronize()
    env.check_errors()
    if rank == 0:
        q.put((full.cpu().numpy(), bool(torch.equal(full, raw)) and bool(torch.equal(full, obs2)), scr.cpu().numpy(), status.cpu().numpy(),
               rew.cpu().numpy(), done.cpu().numpy(), flags.cpu().numpy()))
    dist.barrier()
    env.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("name,kind,n_total", [("nohide", "symbol", 192), ("default", "gray", 250)])
def test_world2_hip_shards_gather_to_the_single_process_batch(name, kind, n_total):
    import torch
    import torch.multiprocessing as mp
    from rogue_gym.envs import DungeonType, HipVecRogueEnv, ImageSetting, StatusFlag

    steps, world = 35, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, kind, n_total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, same, scr, status, rew, done, flags = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same, "compact gather + HIP expand differs from the raw f32 gather"
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))["configs"][name]
    st = ImageSetting(DungeonType.SYMBOL if kind == "symbol" else DungeonType.GRAY, StatusFlag.FULL, kind != "symbol")
    env = HipVecRogueEnv([dict(cfg, seed=i) for i in range(n_total)], max_steps=40, image_setting=st, device=0)
    g = torch.Generator(device="cpu").manual_seed(5)
    for _ in range(steps):
        obs, _, _ = env.step(torch.randint(0, 11, (n_total,), generator=g).to(env.device))
    torch.cuda.synchronize()
    assert np.array_equal(full, obs.cpu().numpy())
    assert np.array_equal(scr, env.screen.cpu().numpy()) and np.array_equal(status, env.status.cpu().numpy())
    # reward / done / message flags of every env of the job arrived with the records (thread_impls.rs:61-81; parallel.py:59-64)
    assert np.array_equal(rew, env.reward.cpu().numpy()) and np.array_equal(done, env.done.cpu().numpy())
    public = 0x1 | 0x2 | 0x7f00 | 0xff0000
    assert np.array_equal(flags, env.flags.cpu().numpy() & public) and (flags != 0).any()
    env.close()


@pytest.mark.timeout(600)
def test_bench_two_ranks_on_one_device():
    """bench.py's N = 2 flow (sharding by rank, barrier, max over ranks, the gather leg) on the HIP stepper; one JSON line from rank 0."""
    env = dict(os.environ, ROGUE_GYM_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    # the driver's own command form: no launcher -- bench.py starts its N ranks itself
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5", "--envs-per-gpu", "8192", "--gather-steps", "10",
           "--clock-warm-s", "0.2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 30 and out["value"] > 0
    assert out["config"]["envs_per_gpu"] == 8192 and "allgather" in out and out["allgather"]["value"] > 0, out.get("allgather")
    # both readings of BASELINE.json's metric in one line: `value` = 8192 envs per GPU here (weak), `strong_scaling` = 65 536 envs in total, split
    ss = out["strong_scaling"]
    assert ss["scaling"] == "strong" and ss["envs_total"] == 65536 and ss["envs_per_gpu"] == 32768 and ss["value"] > 0, ss
    assert out["rccl_saw_n_ranks"] is False   # (the one-device stand-in has no RCCL communicator: the field says so)
    assert out["step_us"]["step"]["p50"] > 0 and out["step_us"]["step"]["max"] >= out["step_us"]["step"]["p99"] >= out["step_us"]["step"]["p50"], out["step_us"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_gpus2_one_device.json"), "w") as f:
        f.write(line + "\n")


@pytest.mark.timeout(900)
def test_bench_eight_ranks_on_one_device():
    """The driver's `python bench.py --gpus 8` is the first run of an 8-process rendezvous: bench.py starts its own 8 ranks
    (ThreadConductor owns its workers the same way, python/src/thread_impls.rs:14-34), gloo control plane on 127.0.0.1, every rank steps its shard, the
    gather legs run, rank 0 prints the one line -- here with all eight ranks on GPU 0 (ROGUE_GYM_BENCH_ONE_DEVICE), small shards."""
    env = dict(os.environ, ROGUE_GYM_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "10", "--warmup", "2", "--envs-per-gpu", "2048", "--gather-steps", "4",
           "--clock-warm-s", "0", "--preroll-steps", "20", "--no-repeats"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["steps"] == 10 and out["value"] > 0
    assert out["config"]["envs_per_gpu"] == 2048 and out["allgather"]["value"] > 0, out.get("allgather")


@pytest.mark.timeout(600)
def test_bench_under_torchrun_still_works():
    """... and the launcher form of the contract (`python -m torch.distributed.run ... bench.py --gpus N`): ranks from the environment."""
    env = dict(os.environ, ROGUE_GYM_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--envs-per-gpu", "4096", "--gather-steps", "4", "--clock-warm-s", "0",
           "--preroll-steps", "20", "--no-repeats"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0


def test_rccl_gather_through_the_cabi_world1():
    """rg_comm_init / rg_allgather_compact with a one-rank communicator: RCCL itself runs (dlopen, ncclCommInitRank, ncclAllGather in place on
    the handle's stream), and the gathered batch is this rank's own records.  (Two ranks cannot share one GPU under RCCL; the >= 2-GPU test
    below needs a multi-GPU box.)"""
    import torch
    from rogue_gym.envs import DungeonType, HipVecRogueEnv, ImageSetting, StatusFlag

    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))["configs"]["mini"]
    st = ImageSetting(DungeonType.GRAY, StatusFlag.FULL, True)
    env = HipVecRogueEnv([dict(cfg, seed=i) for i in range(300)], max_steps=50, image_setting=st, device=0)
    env.init_comm(rank=0, world=1)
    g = torch.Generator(device="cpu").manual_seed(3)
    for _ in range(40):
        env.step(torch.randint(0, 11, (300,), generator=g).to(env.device))
    for with_hist in (False, True):
        got = env.all_gather_records(with_hist).clone()
        assert torch.equal(got, env.packed_records(with_hist))
    assert torch.equal(env.all_gather_obs(compact=True), env.obs)
    torch.cuda.synchronize()
    env.check_errors()
    env.close()


def _nccl_worker(rank, world, port, n_per, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "rogue-gym_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from rogue_gym.envs import DungeonType, HipVecRogueEnv, ImageSetting, StatusFlag

    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))["configs"]["mini"]
    env = HipVecRogueEnv([dict(cfg, seed=rank * n_per + i) for i in range(n_per)], max_steps=40,
                         image_setting=ImageSetting(DungeonType.GRAY, StatusFlag.FULL, False), device=rank)
    g = torch.Generator(device="cpu").manual_seed(11)
    for _ in range(30):
        a = torch.randint(0, 11, (world * n_per,), generator=g)
        env.step(a[rank * n_per:(rank + 1) * n_per].to(env.device))
    via_torch = env.all_gather_obs(compact=True).clone()   # torch.distributed all_gather_into_tensor over RCCL
    raw = env.all_gather_obs(compact=False)
    env.init_comm()                                         # the handle's own communicator
    via_cabi = env.all_gather_obs(compact=True)
    torch.cuda.synchronize()
    ok = bool(torch.equal(via_torch, via_cabi)) and bool(torch.equal(via_torch, raw))
    mine = bool(torch.equal(via_cabi[rank * n_per:(rank + 1) * n_per], env.obs))
    q.put((rank, ok, mine))
    dist.barrier()
    env.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_gather_two_gpus():
    """The real thing: two ranks on two GPUs, backend nccl (= RCCL over xGMI).  Skipped on a 1-GPU box."""
    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, 500, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)]
