#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched HIP Rogue-Gym stepper (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one lock-step pass of the hot path over every env of this rank: action fetch ->
k_step (player turn, monster AI, combat, FoV, descents, auto-reset) -> k_obs (mirror refresh fused with the
f32 observation encode into a PyTorch-ROCm tensor in HBM); k_regen refills the consumed spare levels on a side stream
(one dungeon generation per reset, overlapped; torch.cuda.synchronize() at the end of the timed region waits for it too).
Workload = BASELINE.json configs[1]: data/config-mini.json (32x16, 2x2 rooms, 26 monsters, hidden
dungeon), 65 536 envs per GPU, per-env seed = global env index, uniform-random policy over the 11
RL actions (device-side generator seeded 0), max_steps = 1000, gray-image observation.
Envs shard across ranks in contiguous blocks with no data-path collective (weak scaling: 65 536
envs per GPU); the optional observation all-gather of the north-star text is timed separately and
reported under "allgather".

Rank 0 prints ONE JSON line: the contract fields + "roofline" (dominant kernel, HIP-event timed on
the launch stream) + "cpu_baseline" (the C oracle on the host cores, bounded sample; N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rogue-gym_amd"))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
KERNELS = ["k_step", "k_render", "k_obs", "k_build"]
# per-kernel share of the algorithmic bytes (DESIGN.md "Kernels"): scalars/entities | tile state | obs write (+ mirror read)
KERNEL_ALGO_BYTES = {"k_step": 128, "k_render": 1024, "k_obs": 3072}  # k_obs = fused mirror refresh + encode: 1024 tile + 2048 obs


# --workload: the default is BASELINE.json configs[1] (the one the metric is quoted on).  The others are the larger parity configs of
# SURVEY.md 8(d), benchable through the same harness for DESIGN.md's tables; they are not the headline line.
WORKLOADS = {
    #  name            golden config, envs/GPU, obs kind, algorithmic B/env-step, per-kernel {k_step, k_obs} bytes, description
    "mini":          ("mini", 65536, "gray", 3200, (128, 3072), "config-mini.json 32x16"),
    "default":       ("default", 32768, "gray", 11776, (256, 11520), "config-default.json 80x24 multi-level"),
    "nohide-symbol": ("nohide", 32768, "symbol", 334336, (256, 334080), "config-nohide.json 80x24, one-hot symbol obs [N,43,24,80]"),
}


def golden_config(name):
    with open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")) as f:
        return json.load(f)["configs"][name]


def cpu_baseline(cfg, desc, budget_s=12.0):
    """The C oracle (a port, not the Rust reference) on all host cores, same workload shape."""
    import numpy as np
    from oracle.pyoracle import OracleBatch

    cores = os.cpu_count() or 1
    n = 65536 if cores >= 64 else 8192  # enough envs per thread that the per-step barrier is noise
    b = OracleBatch([cfg] * n, max_steps=1000, n_threads=cores, seeds=list(range(n)))
    obs = np.zeros((n, 1, cfg["height"], cfg["width"]), np.float32)
    acts = np.frombuffer(b".hjklnbuy>s", np.uint8)
    rng = np.random.RandomState(0)
    keys = [acts[rng.randint(0, 11, n)] for _ in range(32)]
    for t in range(5):
        b.step(keys[t], obs)
    steps, t0 = 0, time.time()
    while time.time() - t0 < budget_s:
        for t in range(10):
            b.step(keys[(steps + t) % 32], obs)
        steps += 10
    dt = time.time() - t0
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d lock-step steps of %s (seeds 0..%d, random 11-action policy, gray obs), C oracle -O3, %d pthreads, %.1f s"
                      % (n, steps, desc, n - 1, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="mini")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="default: the workload's own size (65 536 for mini)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--time-every", type=int, default=8, help="bracket every N-th kernel launch with HIP events (roofline leg)")
    ap.add_argument("--gather-steps", type=int, default=50, help="extra steps timed WITH the observation all-gather (N>1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # development aid: ROGUE_GYM_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 with the gloo backend, so the multi-rank control flow
    # (sharding, barrier, max-over-ranks, gather) can be exercised on a 1-GPU box.  Never set by the driver.
    one_device = os.environ.get("ROGUE_GYM_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from rogue_gym.envs import DungeonType, HipVecRogueEnv, ImageSetting, StatusFlag

    cfg_name, n_default, obs_kind, algo_bytes, (step_bytes, obs_bytes), wl_desc = WORKLOADS[args.workload]
    cfg = golden_config(cfg_name)
    n = args.envs_per_gpu or n_default
    KERNEL_ALGO_BYTES.update({"k_step": step_bytes, "k_obs": obs_bytes})
    first = rank * n
    cfgs = [json.dumps(dict(cfg, seed=first + i)) for i in range(n)]
    env = HipVecRogueEnv(cfgs, max_steps=1000, image_setting=ImageSetting(DungeonType.GRAY if obs_kind == "gray" else DungeonType.SYMBOL, StatusFlag.EMPTY, False),
                         device=local_rank)
    L, h = env._h.L, env._h.h

    K, W = args.steps, args.warmup
    gen = torch.Generator(device=dev).manual_seed(rank)  # action source is not part of parity
    # pre-generated action tensor; the per-step "action fetch" is the index -> key gather below
    chunk = 256
    actions = torch.randint(0, 11, (min(chunk, K + W), n), generator=gen, device=dev, dtype=torch.int64)
    keys_all = env._action_keys[actions].contiguous()

    def one_step(t):
        return env.step_keys(keys_all[t % keys_all.shape[0]])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for t in range(W):
        one_step(t)
    env._h.check(L.rg_timing_enable(h, args.time_every))  # HIP-event pairs on the launch stream around every N-th launch
    barrier()
    t0 = time.perf_counter()
    for t in range(K):
        one_step(W + t)
    barrier()
    dt = time.perf_counter() - t0
    ms = (C.c_double * 4)()
    cnt = (C.c_uint64 * 4)()
    env._h.check(L.rg_timing_read(h, ms, cnt))
    env._h.check(L.rg_timing_enable(h, 0))
    env.check_errors()

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    # optional: the north-star's observation all-gather (compact u8 screen + i32 status), timed separately
    gather = None
    if world > 1 and args.gather_steps > 0:
        for t in range(5):
            one_step(t); env.all_gather_obs(compact=True)
        barrier()
        g0 = time.perf_counter()
        for t in range(args.gather_steps):
            one_step(t); env.all_gather_obs(compact=True)
        barrier()
        gdt = torch.tensor([time.perf_counter() - g0], dtype=torch.float64, device=dev)
        dist.all_reduce(gdt, op=dist.ReduceOp.MAX)
        gather = {"value": n * world * args.gather_steps / float(gdt.item()), "unit": "env-steps/s",
                  "payload": "u8 screen [N,16,32] + i32 status [N,10] all-gathered to every rank each step (RCCL)"}

    if rank == 0:
        per_kernel = {}
        for k in range(3):
            if cnt[k]:
                avg_ms = ms[k] / cnt[k]
                per_kernel[KERNELS[k]] = {"avg_us": avg_ms * 1e3, "launches": int(cnt[k]),
                                          "algo_GBps": KERNEL_ALGO_BYTES[KERNELS[k]] * n / (avg_ms * 1e-3) / 1e9}
        dom = max(per_kernel, key=lambda k: per_kernel[k]["avg_us"]) if per_kernel else "k_step"
        dom_s = per_kernel[dom]["avg_us"] * 1e-6 if per_kernel else dt_max / K
        achieved = algo_bytes * n / dom_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written from rocprofv3 --pmc passes (see profiles/README.md)
        if os.path.exists(pmc) and args.workload == "mini" and n == 65536:  # the PMC passes ran on 65 536 envs per launch
            with open(pmc) as f:
                traffic = json.load(f).get(dom, {}).get("hbm_bytes_per_launch")
        out = {
            "metric": "env-steps/sec (whole node) at 65 536 envs, 32x16 mini-dungeon" if args.workload == "mini" and n == 65536
                      else "env-steps/sec (whole node), workload %s, %d envs per GPU" % (args.workload, n),
            "value": n * world * K / dt_max,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt_max / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/u16 integer state, f32 observation",
            "data": "synthetic (per-env seed = env index, uniform-random 11-action policy)",
            "config": {"workload": "%s, %d envs per GPU (%d total), %s-image obs [N,%d,%d,%d] f32, max_steps 1000, auto-reset"
                                   % (wl_desc, n, n * world, obs_kind, env.channels, env.height, env.width), "envs_per_gpu": n, "parallelism": "env-sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic,
                         "note": "achieved = %d algorithmic B/env-step x %d envs / avg %s duration (HIP events on the launch stream); "
                                 "the step kernel is latency/divergence-bound, not bandwidth-bound" % (algo_bytes, n, dom),
                         "per_kernel": per_kernel},
        }
        if gather:
            out["allgather"] = gather
        if world == 1 and not args.no_cpu_baseline and obs_kind == "gray":
            out["cpu_baseline"] = cpu_baseline(cfg, wl_desc)
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
