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
    torch.cuda.synchronize()
    env.check_errors()
    if rank == 0:
        q.put((full.cpu().numpy(), bool(torch.equal(full, raw)), scr.cpu().numpy(), status.cpu().numpy()))
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
    full, same, scr, status = q.get(timeout=240)
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
    env.close()


@pytest.mark.timeout(600)
def test_bench_two_ranks_on_one_device():
    """bench.py's N = 2 flow (sharding by rank, barrier, max over ranks, the gather leg) on the HIP stepper; one JSON line from rank 0."""
    env = dict(os.environ, ROGUE_GYM_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5", "--envs-per-gpu", "8192", "--gather-steps", "10", "--clock-warm-s", "0.2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 30 and out["value"] > 0
    assert out["config"]["envs_per_gpu"] == 8192 and "allgather" in out and out["allgather"]["value"] > 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_gpus2_one_device.json"), "w") as f:
        f.write(line + "\n")
