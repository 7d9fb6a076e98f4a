#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched HIP Rogue-Gym stepper (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its own N ranks, one per GPU, RCCL)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (ranks from the launcher's environment)

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

Timed region: W untimed warm-up steps, barrier + synchronize, EXACTLY K steps, barrier + synchronize; max over ranks.
Before the warm-up there is a DISCLOSED, fixed-duration clock-warm phase ("clock_warm" in the JSON): a scratch batch of
envs (not the measured one) is stepped for --clock-warm-s seconds so that a cold GPU has ramped its shader clock before
anything is timed -- `steps` / `warmup` keep their meaning.  The effective shader clock is probed (s_memtime vs the
100 MHz s_memrealtime) before and after and reported.
The metric is defined on the STEADY-STATE episode mix ("steady-state over >= 1000 lock-step batch steps after warm-up,
including auto-resets"): right after creation all 65 536 envs are at step 0 of their first episode -- every monster of the start room wakes
at once, every DistCache is cold, nobody resets -- a heavier, unrepresentative transient (measured: k_step 160 us vs 100 us; clocks and spares
have nothing to do with it, see profiles/r02_driver_repro.txt).  A DISCLOSED pre-roll ("preroll" in the JSON: --preroll-steps untimed steps
of the measured batch, default 1500 = 1.5 max_steps horizons -- at step 1000 the 40 % of the envs that survived their first episode hit max_steps together
and the next few dozen steps are a second such transient, so the window must not sit on a multiple of max_steps) therefore precedes the W warm-up steps; the first K steps of that pre-roll are
timed too and reported as "preroll.cold_start" so that the transient's own rate is on record next to `value`.

Rank 0 prints ONE JSON line: the contract fields + "roofline" (dominant kernel, HIP-event timed on the launch stream,
plus the end-to-end and per-kernel fractions and the on-box copy peak) + "workload_rates" (resets/s, descents/s,
BFS maps/s) + "repeats" (4 further runs of K steps, median) + "extra_workloads" (the 80x24 default and nohide-symbol
configs of SURVEY.md 8d through the same harness, short runs) + "cpu_baseline" (the C oracle on the host cores,
bounded samples at 65 536 / 1 024 / 64 envs; N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rogue-gym_amd"))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
KERNELS = ["k_step", "k_render", "k_obs", "k_build", "k_regen"]   # rg_timing_read_all's order; k_regen = the background generator (side stream)

# --workload: the default is BASELINE.json configs[1] (the one the metric is quoted on).  The others are the larger configs of
# SURVEY.md 8(d); the default run also reports them (short) under "extra_workloads".
WORKLOADS = {
    #  name            golden config, envs/GPU, obs kind, algorithmic B/env-step, per-kernel {k_step, k_obs} bytes, description
    "mini":          ("mini", 65536, "gray", 3200, (128, 3072), "config-mini.json 32x16"),
    "default":       ("default", 32768, "gray", 11776, (256, 11520), "config-default.json 80x24 multi-level"),
    "nohide-symbol": ("nohide", 32768, "symbol", 334336, (256, 334080), "config-nohide.json 80x24, one-hot symbol obs [N,43,24,80]"),
}


def golden_config(name):
    with open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")) as f:
        return json.load(f)["configs"][name]


_ORACLE_BUILD = None


def oracle_build():
    """The baseline legs time the C restatement compiled ON THIS HOST with -O3 -march=native; without a compiler here, the
    portable -O3 library the tests use."""
    global _ORACLE_BUILD
    if _ORACLE_BUILD is None:
        from oracle import pyoracle
        _ORACLE_BUILD = "-O3 -march=native, built on this host" if pyoracle.build_native() else "-O3 (portable build: no compiler on this host)"
    return _ORACLE_BUILD


def cpu_sample(cfg, n, cores, budget_s):
    import numpy as np
    oracle_build()
    from oracle.pyoracle import OracleBatch

    b = OracleBatch([cfg] * n, max_steps=1000, n_threads=min(cores, n), seeds=list(range(n)))
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
    return n * steps / dt, steps, dt, min(cores, n)


def cpu_baseline(cfg, desc, budget_s=10.0):
    """The C oracle (a port, not the Rust reference) on all host cores, same workload shape."""
    cores = os.cpu_count() or 1
    n = 65536 if cores >= 64 else 8192  # enough envs per thread that the per-step barrier is noise
    v, steps, dt, used = cpu_sample(cfg, n, cores, budget_s)
    out = {"value": v, "unit": "env-steps/s", "cores": used, "kind": "port", "build": oracle_build(),
           "sample": "%d envs x %d lock-step steps of %s (seeds 0..%d, random 11-action policy, gray obs), C oracle, %d pthreads, %.1f s"
                     % (n, steps, desc, n - 1, used, dt), "other_sizes": {},
           "note": "varies by box and run (3.8 - 4.6 M on the 256-core hosts of this pool); a reported baseline, not the target"}
    for m in (1024, 64):  # the CPU side at 64 / 1 024 envs too (what the reference's thread-per-env design can reach)
        v2, s2, dt2, used2 = cpu_sample(cfg, m, cores, 2.0)
        out["other_sizes"][str(m)] = {"value": v2, "cores": used2, "sample": "%d envs x %d steps, %.1f s" % (m, s2, dt2)}
    return out


class Harness:
    """One workload on this rank: env batch + pre-generated action tensor."""

    def __init__(self, torch, workload, n, rank, local_rank, persistent_obs=False):
        from rogue_gym.envs import DungeonType, HipVecRogueEnv, ImageSetting, StatusFlag

        cfg_name, n_default, obs_kind, self.algo_bytes, (self.step_bytes, self.obs_bytes), self.desc = WORKLOADS[workload]
        self.cfg = golden_config(cfg_name)
        self.n = n or n_default
        self.obs_kind = obs_kind
        first = rank * self.n
        cfgs = [json.dumps(dict(self.cfg, seed=first + i)) for i in range(self.n)]
        self.env = HipVecRogueEnv(cfgs, max_steps=1000,
                                  image_setting=ImageSetting(DungeonType.GRAY if obs_kind == "gray" else DungeonType.SYMBOL, StatusFlag.EMPTY, False),
                                  device=local_rank, persistent_obs=persistent_obs)
        dev = self.env.device
        gen = torch.Generator(device=dev).manual_seed(rank)  # action source is not part of parity
        # pre-generated action tensor (512 rows whatever --steps is: a table of only steps + warmup rows -- 25 for the driver's command -- turns the
        # "uniform-random policy" into a 25-periodic one when it is cycled over a 1500-step pre-roll, a different and heavier workload; that, not
        # clocks or spares, was the rest of the gap between the round-1 driver line and the 2000-step run); the per-step "action fetch" is the row lookup
        rows = 512
        actions = torch.randint(0, 11, (rows, self.n), generator=gen, device=dev, dtype=torch.uint8)
        self.keys_all = self.env._action_keys[actions.long()].contiguous()
        self.t = 0

    def step(self):
        r = self.env.step_keys(self.keys_all[self.t % self.keys_all.shape[0]])
        self.t += 1
        return r

    def timing(self, every):
        self.env._h.check(self.env._h.L.rg_timing_enable(self.env._h.h, every))

    def read_timing(self):
        """Per kernel: average duration over the SAMPLED launches (HIP events stamped with the dispatch's own begin / end), how many launches there were in
        all, and `duration_share` = avg x launches / sum over the kernels -- who holds the GPU for how long (k_regen runs BESIDE k_step on a low-priority
        stream, so the shares are not a partition of wall time)."""
        nk = len(KERNELS)
        ms, cnt, tot = (C.c_double * nk)(), (C.c_uint64 * nk)(), (C.c_uint64 * nk)()
        self.env._h.check(self.env._h.L.rg_timing_read_all(self.env._h.h, nk, ms, cnt, tot))
        out = {}
        kb = {"k_step": self.step_bytes, "k_obs": self.obs_bytes, "k_render": 1024}
        for k in range(nk):
            if cnt[k]:
                avg_ms = ms[k] / cnt[k]
                out[KERNELS[k]] = {"avg_us": avg_ms * 1e3, "sampled": int(cnt[k]), "launches": int(tot[k])}
                if KERNELS[k] in kb:
                    gbps = kb[KERNELS[k]] * self.n / (avg_ms * 1e-3) / 1e9
                    out[KERNELS[k]].update({"algo_bytes_per_env": kb[KERNELS[k]], "algo_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS})
        total = sum(v["avg_us"] * v["launches"] for v in out.values()) or 1.0
        for v in out.values():
            v["duration_share"] = v["avg_us"] * v["launches"] / total
        return out

    def step_percentiles(self, steps):
        """`steps` steps with an event pair on EVERY launch (their own window: the pairs cost stream time, so never the timed region): percentiles of the GPU
        time of a step -- begin of its k_step to end of its observation pass -- and of the two kernels, in microseconds.  What a learner that waits on every
        step sees, next to the mean the headline is (the level-per-lane producer's rounds show up here, not in the mean)."""
        import numpy as np
        steps = min(steps, 4000)
        self.timing(1)
        for _ in range(steps):
            self.step()
        out = {"steps": steps}
        buf, got = (C.c_float * 4096)(), C.c_int()
        for name, k in (("step", -1), ("k_step", 0), ("k_obs", 2)):
            self.env._h.check(self.env._h.L.rg_timing_read_samples(self.env._h.h, k, buf, 4096, C.byref(got)))
            if got.value:
                a = np.sort(np.frombuffer(buf, np.float32, got.value).astype(np.float64)) * 1e3
                out[name] = {"p50": float(a[len(a) // 2]), "p90": float(a[int(len(a) * 0.9)]), "p99": float(a[int(len(a) * 0.99)]), "max": float(a[-1]), "mean": float(a.mean()),
                             "samples": int(got.value)}
        self.timing(0)
        if "step" in out:
            out["p99_over_p50"] = out["step"]["p99"] / out["step"]["p50"]
        return out

    def sclk(self):
        mhz = C.c_double()
        self.env._h.check(self.env._h.L.rg_probe_sclk(self.env._h.h, C.byref(mhz)))
        return mhz.value

    def close(self):
        self.env.close()


def copy_peak(torch, dev):
    """On-box device-to-device copy rate (read + write bytes per second), the achievable-HBM reference next to the 8 TB/s nominal figure."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 20
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    del a, b
    return 2.0 * n * reps / (ms * 1e-3) / 1e9


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (the reference's executor owns its workers too,
    python/src/thread_impls.rs:14-34) -- one process per GPU, rendezvous on 127.0.0.1, RCCL.  Rank 0 inherits stdout and prints the one
    JSON line; the exit code is the worst of the ranks'."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, live = 0, list(procs)
    try:
        while live:  # one rank failing must not leave the others blocked in a collective (and this process waiting on them) forever
            for p in list(live):
                r = p.poll()
                if r is not None:
                    live.remove(p)
                    rc = max(rc, abs(r))
            if rc:
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="mini")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="default: the workload's own size (65 536 for mini)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--persistent-obs", action="store_true", help="A/B aid: the measured batch with its observation tensor bound (rg_obs_bind: in-place updates); the line then "
                    "says so in `config.workload` and is not the headline metric")
    ap.add_argument("--no-extra", action="store_true", help="skip the short extra_workloads runs (default / nohide-symbol)")
    ap.add_argument("--no-repeats", action="store_true", help="skip the 4 further runs of K steps (median)")
    ap.add_argument("--clock-warm-s", type=float, default=1.5, help="untimed fixed-duration stepping of a SCRATCH batch before anything else (0 = off)")
    ap.add_argument("--preroll-steps", type=int, default=1500, help="untimed steps of the MEASURED batch before the warm-up: reach the steady-state episode mix "
                    "the metric is defined on (the first --steps of them are timed and reported as preroll.cold_start); 0 = off.  1.5 x max_steps: "
                    "not on a multiple of max_steps, where the survivors of the synchronised first episodes all reset at once")
    ap.add_argument("--time-every", type=int, default=0, help="time every N-th launch of each kernel with a HIP-event pair (default: every 7th; every 5th when steps < 64, every one when < 16); odd on purpose: "
                    "an even stride could lock onto a period of the workload")
    ap.add_argument("--gather-steps", type=int, default=50, help="extra steps timed WITH the observation all-gather (N>1)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))

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
    elif torch.cuda.device_count() < min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        sys.exit("bench.py: %d ranks on this node but only %d GPU(s) visible (one rank per GPU; ROGUE_GYM_BENCH_ONE_DEVICE=1 is the 1-GPU development mode)"
                 % (world, torch.cuda.device_count()))
    data_group, data_group_error = None, None
    if world > 1:
        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        # Two process groups.  CONTROL (barrier, max over ranks of the wall time: CPU scalars) is gloo over 127.0.0.1 -- the sharded path has no
        # data-path collective, so nothing about the GPU interconnect may take the headline down with it.  DATA is backend "nccl" =
        # RCCL over xGMI: the all-gather legs below run on it (its communicator is created at the first collective, inside the guarded leg).
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        if not one_device:
            try:
                data_group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
            except Exception as e:  # noqa: BLE001
                data_group_error = "%s: %s" % (type(e).__name__, e)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    K, W = args.steps, args.warmup

    # ---- disclosed clock-warm phase on a scratch batch (never the measured one) ----
    clock_warm = None
    if args.clock_warm_s > 0:
        scratch = Harness(torch, "mini", 16384, rank, local_rank)
        mhz0 = scratch.sclk()
        t0, warm_steps = time.perf_counter(), 0
        while time.perf_counter() - t0 < args.clock_warm_s:
            for _ in range(50):
                scratch.step()
            torch.cuda.synchronize()
            warm_steps += 50
        mhz1 = scratch.sclk()
        clock_warm = {"seconds": args.clock_warm_s, "batch": "scratch batch of 16384 mini envs (destroyed before the measured batch is created)",
                      "scratch_steps": warm_steps, "sclk_mhz_before": mhz0, "sclk_mhz_after": mhz1}
        scratch.close()
        del scratch

    hz = Harness(torch, args.workload, args.envs_per_gpu, rank, local_rank, persistent_obs=args.persistent_obs)
    n, env = hz.n, hz.env
    preroll = None
    if args.preroll_steps > 0:
        kc = min(K, args.preroll_steps)
        barrier()
        c0 = time.perf_counter()
        for _ in range(kc):
            hz.step()
        barrier()
        cdt = torch.tensor([time.perf_counter() - c0], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(cdt, op=dist.ReduceOp.MAX)
        for _ in range(args.preroll_steps - kc):
            hz.step()
        torch.cuda.synchronize()
        preroll = {"steps": args.preroll_steps,
                   "why": "the metric is the steady-state rate (SURVEY.md 8d); a freshly created batch has every env at step 0 of its first episode "
                          "(all start-room monsters awake, cold DistCaches, no resets), a heavier transient",
                   "cold_start": {"steps": kc, "ms_per_step": float(cdt.item()) / kc * 1e3, "value": n * world * kc / float(cdt.item()),
                                  "note": "the first %d steps after creation, no warm-up at all" % kc}}
    hz.timing(1); hz.timing(0)  # create the event pool NOW (32 768 hipEventCreate calls: ~50 ms of host time with an idle GPU) -- not between the warm-up and the timed region
    for _ in range(W):
        hz.step()
    # HIP-event pairs handed to a launch cost stream time: measured in round 4 at ~10 us per step with every 3rd launch of both kernels timed (a
    # 20-step window: 131-143 us per step against 120-129 without, tools/tmp/window_fit2.py).  So few samples for short runs: every 5th launch = 4 pairs
    # per kernel for the driver's 20 steps (round 5, same box: 651-656 M with the four pairs per kernel, 666-676 M with none; two pairs cost the same as four).  The strides are odd out of caution only: since round 4 the generator runs beside EVERY k_step, there are no
    # two kinds of launch left to alternate between.
    every = args.time_every if args.time_every > 0 else (1 if K < 16 else (5 if K < 64 else 7))
    hz.timing(every)  # HIP-event pairs on the launch stream around every `every`-th launch of each kernel
    env.counters(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        hz.step()
    barrier()
    dt = time.perf_counter() - t0
    per_kernel = hz.read_timing()
    hz.timing(0)
    counts = env.counters(reset=True)
    sclk_after = hz.sclk()
    env.check_errors()

    tmax = torch.tensor([dt], dtype=torch.float64)
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": dev.index, "pci_bus_id": getattr(props, "pci_bus_id", None),
            "name": props.name}
    rank_devices = [mine]
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)
    dt_max = float(tmax.item())

    # ---- 4 further runs of K steps (no event timing): the spread of the headline number ----
    repeats = None
    if not args.no_repeats:
        runs = [dt_max / K * 1e3]
        for _ in range(4):
            barrier()
            r0 = time.perf_counter()
            for _ in range(K):
                hz.step()
            barrier()
            r = torch.tensor([time.perf_counter() - r0], dtype=torch.float64)
            if world > 1:
                dist.all_reduce(r, op=dist.ReduceOp.MAX)
            runs.append(float(r.item()) / K * 1e3)
        med = statistics.median(runs)
        repeats = {"ms_per_step": runs, "median_ms_per_step": med, "median_value": n * world / (med * 1e-3),
                   "note": "run 0 is the timed region `value` is computed from; runs 1-4 repeat it without HIP-event bracketing"}

    # SURVEY.md 8(d) defines the metric on "steady-state over >= 1 000 lock-step batch steps": when the timed region is shorter (the driver's command:
    # 20 steps), the same batch runs one 1 000-step window right behind it, so that the contract's own window is in the line too
    long_window = None
    if K < 1000:
        barrier()
        l0 = time.perf_counter()
        for _ in range(1000):
            hz.step()
        barrier()
        lw = torch.tensor([time.perf_counter() - l0], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(lw, op=dist.ReduceOp.MAX)
        long_window = {"steps": 1000, "ms_per_step": float(lw.item()) / 1000 * 1e3, "value": n * world * 1000 / float(lw.item())}

    # what one step costs a caller that waits on every step: percentiles over their own 1 000-step window (event pair on every launch)
    step_us = hz.step_percentiles(1000)

    # BASELINE.json's metric reads "whole node at 65 536 envs ... 1/2/4/8 GPU": `value` keeps 65 536 envs PER GPU (weak scaling, the contract's default); this
    # leg is the other reading -- 65 536 envs IN TOTAL, 65 536 / N per GPU -- so that the line answers both.  A second, smaller batch on every rank (seeds = the
    # global env index again), its own pre-roll, 200 timed steps, max over ranks.
    strong = None
    if world > 1:
        total = 65536 if args.workload == "mini" else WORKLOADS[args.workload][1]
        per = max(64, total // world)
        sz = Harness(torch, args.workload, per, rank, local_rank)
        for _ in range(600 + 20):
            sz.step()
        barrier()
        s0 = time.perf_counter()
        for _ in range(200):
            sz.step()
        barrier()
        sdt = torch.tensor([time.perf_counter() - s0], dtype=torch.float64)
        dist.all_reduce(sdt, op=dist.ReduceOp.MAX)
        sz.env.check_errors()
        sz.close()
        del sz
        strong = {"scaling": "strong", "envs_total": per * world, "envs_per_gpu": per, "steps": 200, "preroll": 600, "warmup": 20, "ms_per_step": float(sdt.item()) / 200 * 1e3,
                  "value": per * world * 200 / float(sdt.item()), "unit": "env-steps/s",
                  "note": "the same workload with the batch SPLIT over the GPUs; k_step is a per-wave latency chain, so a smaller shard is not proportionally faster (DESIGN.md section 6)"}

    # optional: the north-star's observation all-gather (ONE collective of the packed compact records, expanded by HIP kernels on the consumer).
    # Two legs: through torch.distributed (all_gather_into_tensor, backend nccl = RCCL) and through the C-ABI's own communicator
    # (rg_comm_init / rg_allgather_compact: pack into the rank's slice + ncclAllGather in place on the handle's stream).  Neither can take the
    # headline down: each leg runs under a watchdog (a collective that never completes cannot be caught as an exception), and a failure is
    # reported inside "allgather" instead of being raised.
    gather = None
    if world > 1 and args.gather_steps > 0:
        import threading

        rec = env._h.L.rg_compact_record_bytes(env._h.h, 0)
        payload = ("one all-gather per step of %d-byte records (u8 screen [%d,%d] + i32 status [10] + f32 reward + u32 flags) = %.1f MB per rank, expanded to f32 [N,%d,%d,%d] "
                   "on every rank by rg_expand_compact" % (rec, env.height, env.width, rec * n / 1e6, env.channels, env.height, env.width))
        gather = {"unit": "env-steps/s", "payload": payload,
                  "process_groups": {"control": "gloo: barrier + max over ranks (CPU scalars)",
                                     "data": "gloo (ROGUE_GYM_BENCH_ONE_DEVICE)" if one_device else ("nccl = RCCL: every all-gather below" if data_group is not None else "unavailable: %s" % data_group_error)}}
        env.process_group = data_group  # torch.distributed leg: all_gather_into_tensor on the RCCL group

        def leg(name):
            for _ in range(5):
                hz.step(); env.all_gather_obs(compact=True)
            barrier()
            g0 = time.perf_counter()
            for _ in range(args.gather_steps):
                hz.step(); env.all_gather_obs(compact=True)
            barrier()
            gdt = torch.tensor([time.perf_counter() - g0], dtype=torch.float64)
            dist.all_reduce(gdt, op=dist.ReduceOp.MAX)
            gather[name] = n * world * args.gather_steps / float(gdt.item())

        def guarded(name, fn, limit_s=120.0):
            done = threading.Event()

            def bail():  # the leg hangs: rank 0 still owes the driver its one JSON line
                if not done.is_set():
                    gather[name + "_error"] = "no completion within %.0f s" % limit_s
                    if rank == 0 and emit_line is not None:
                        emit_line()
                    os._exit(0)

            t = threading.Timer(limit_s, bail)
            t.daemon = True
            t.start()
            try:
                fn()
            except Exception as e:  # noqa: BLE001
                gather[name + "_error"] = "%s: %s" % (type(e).__name__, e)
            done.set()
            t.cancel()

        gather_legs = []
        if one_device or data_group is not None:
            gather_legs.append(("value", lambda: leg("value")))   # torch.distributed
        else:
            gather["value_error"] = "no RCCL process group: %s" % data_group_error
        if not one_device:  # two ranks cannot share one GPU under RCCL: the C-ABI communicator needs one device per rank
            def cabi():
                env.init_comm()
                cnt, urank = env.comm_count()   # what RCCL itself says (ncclCommCount / ncclCommUserRank)
                gather["ranks"] = {"rccl_comm_count": cnt, "rccl_user_rank_of_rank0": urank, "world_size": world,
                                   "torch_data_group_size": dist.get_world_size(data_group) if data_group is not None else None}
                if out is not None:
                    out["rccl_saw_n_ranks"] = bool(cnt == world)
                leg("value_cabi")
            gather_legs.append(("value_cabi", cabi))
    else:
        gather_legs = []

    out = None
    if rank == 0:
        # the dominant kernel of the STEP (what bounds ms_per_step): the longer of the two serial kernels.  k_regen is timed too and shown in
        # gpu_time_share -- by duration x launches it holds the GPU as long as k_step does -- but it runs beside k_step on a low-priority stream and
        # moves ~7 MB per launch (SALU-bound level generation): pricing the step's bytes against ITS clock would say nothing about either.
        serial = {k: v for k, v in per_kernel.items() if k in ("k_step", "k_obs", "k_render")}
        dom = max(serial, key=lambda k: serial[k]["avg_us"]) if serial else "k_step"
        dom_s = per_kernel[dom]["avg_us"] * 1e-6 if serial else dt_max / K
        achieved = hz.algo_bytes * n / dom_s / 1e9
        own_bytes = {"k_step": hz.step_bytes, "k_obs": hz.obs_bytes}.get(dom, hz.algo_bytes) * n  # the dominant kernel's OWN algorithmic bytes per launch
        e2e = hz.algo_bytes * n / (dt_max / K) / 1e9
        traffic, traffic_all, traffic_src, traffic_build = None, None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written from rocprofv3 --pmc passes (see profiles/README.md)
        if os.path.exists(pmc) and args.workload == "mini" and n == 65536:  # the PMC passes ran on 65 536 envs per launch
            with open(pmc) as f:
                pj = json.load(f)
            traffic_build = pj.get("_build_id")
            traffic = pj.get(dom, {}).get("hbm_bytes_per_launch")
            traffic_src = pj.get("_measured", "profiles/pmc_traffic.json (rocprofv3 --pmc passes; build not stamped)")
            traffic_all = {k: {"hbm_bytes_per_launch": v.get("hbm_bytes_per_launch"), "hbm_bytes_per_env_step": (v.get("hbm_bytes_per_launch") or 0) / 65536.0}
                           for k, v in pj.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}
        build_id = env._h.L.rg_build_id().decode()
        if traffic is not None and traffic_build != build_id:
            # the PMC passes are a separate rocprofv3 collection (profiles/pmc_traffic.json): a figure taken on another build of the kernels is not reported
            traffic_src = "profiles/pmc_traffic.json was collected on build %s, this library is %s: traffic withheld" % (traffic_build, build_id)
            traffic, traffic_all = None, None
        out = {
            "metric": "env-steps/sec (whole node) at 65 536 envs, 32x16 mini-dungeon" if args.workload == "mini" and n == 65536 and not args.persistent_obs
                      else "env-steps/sec (whole node), workload %s, %d envs per GPU" % (args.workload, n),
            "value": n * world * K / dt_max,
            "unit": "env-steps/s",
            "n_gpus": world,
            "n_gpus_visible": torch.cuda.device_count(),
            "rank_devices": rank_devices,   # per rank: LOCAL_RANK, the HIP device it stepped on and that device's PCI bus id -- a mis-mapped LOCAL_RANK shows here
            "steps": K,
            "warmup": W,
            "ms_per_step": dt_max / K * 1e3,
            "value_is": "steady-state episode mix: EXACTLY `steps` timed steps after a disclosed pre-roll of the measured batch (`preroll`) + `warmup` steps",
            "value_cold_start": preroll["cold_start"]["value"] if preroll else None,   # the first `steps` steps after creation, no warm-up at all
            "value_median_of_repeats": repeats["median_value"] if repeats else None,   # 5 runs of `steps` steps; runs 1-4 without HIP-event pairs
            "value_long_window": long_window["value"] if long_window else None,        # 1 000 steps right behind the timed region (SURVEY.md 8d's own window), when `steps` < 1 000
            "long_window": long_window,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/u16 integer state, f32 observation",
            "data": "synthetic (per-env seed = env index, uniform-random 11-action policy)",
            "config": {"workload": "%s, %d envs per GPU (%d total), %s-image obs [N,%d,%d,%d] f32, max_steps 1000, auto-reset"
                                   % (hz.desc, n, n * world, hz.obs_kind, env.channels, env.height, env.width), "envs_per_gpu": n,
                       "parallelism": "env-sharded x%d, no data-path collective" % world},
            "build_id": build_id,  # sha256[:16] of the library's sources (__graft_entry__.source_id)
            "clock_warm": clock_warm,
            "preroll": preroll,
            "sclk_mhz_after_timed_region": sclk_after,
            # flat and small on purpose (the driver keeps a prefix of the line).  `achieved` is the contract's definition: the WHOLE step's algorithmic
            # bytes over the dominant kernel's clock.  `traffic` is that kernel's own measured HBM bytes per launch (PMC, calibrated), to be read against
            # `kernel_algorithmic_bytes` -- ITS OWN share of the step's bytes -- not against `achieved`: k_step touches 4.7x what it needs (a 64-B line per
            # 2-byte word) and is still at 7 % of HBM: a latency kernel.
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_build_id": traffic_build, "kernel_algorithmic_bytes": own_bytes,
                         "traffic_over_kernel_algorithmic": (traffic / own_bytes) if traffic else None,
                         "kernel_avg_us": dom_s * 1e6, "step_algorithmic_bytes": hz.algo_bytes * n,
                         "frac_end_to_end": e2e / HBM_PEAK_GBPS, "achieved_end_to_end": e2e,
                         "gpu_time_share": {k: round(v["duration_share"], 4) for k, v in per_kernel.items()},
                         "traffic_source": traffic_src},
            "kernels": {"per_kernel": per_kernel, "pmc_traffic": traffic_all,
                        "event_sampling": "every %d-th launch of each kernel carries a HIP-event pair stamped with the dispatch's own begin / end (hipExtLaunchKernelGGL on the launch stream; k_regen on its side stream)" % every,
                        "note": "roofline.achieved = %d algorithmic B/env-step x %d envs / avg %s duration (the contract's definition: it charges the whole step's "
                                "bytes to the dominant kernel); frac_end_to_end = the same bytes / ms_per_step; per_kernel has every kernel's own algorithmic share. "
                                "k_step is latency / instruction-issue / divergence-bound, k_obs is the HBM-side kernel, k_regen (background level generation: spare level-1 "
                                "states and next-level structures, one launch beside every k_step) is scalar-unit-bound." % (hz.algo_bytes, n, dom)},
            "step_us": step_us,
            "strong_scaling": strong,   # N > 1: 65 536 envs in TOTAL (65 536 / N per GPU) beside the weak-scaling `value`
            "rccl_saw_n_ranks": None if world == 1 else False,   # N > 1: set by the C-ABI all-gather leg iff ncclCommCount == WORLD_SIZE
            "workload_rates": {"per": "whole job, per second (rank 0's counters x n_gpus)",
                               **{k + "_per_s": v * world / dt_max for k, v in counts.items()},
                               "per_batch_step": {k: v / K for k, v in counts.items()}},
        }
        if repeats:
            out["repeats"] = repeats
        if gather is not None:
            out["allgather"] = gather  # filled in by the legs below

    emit_line = None
    if rank == 0:
        def emit_line():
            print(json.dumps(out), flush=True)
    for name, fn in gather_legs:
        guarded(name, fn)
    hz.close()
    del hz, env

    if rank == 0 and world == 1:
        try:
            peak = copy_peak(torch, dev)
            out["roofline"]["copy_peak_GBps"] = peak
            out["roofline"]["frac_of_copy_peak"] = out["roofline"]["achieved"] / peak
            for v in out["kernels"]["per_kernel"].values():
                if "algo_GBps" in v:
                    v["frac_of_copy_peak"] = v["algo_GBps"] / peak
        except Exception as e:  # never lose the headline line to the side measurement
            out["roofline"]["copy_peak_GBps"] = None
            out["roofline"]["copy_peak_error"] = str(e)
        if args.workload == "mini" and not args.no_extra:
            extra = {}
            for name, ksteps, kwarm, kpre in (("default", 300, 50, 500), ("nohide-symbol", 60, 10, 100)):
                try:
                    x = Harness(torch, name, 0, 0, local_rank)
                    for _ in range(kpre + kwarm):  # (same reason as the main pre-roll: not the synchronised first steps of every env)
                        x.step()
                    x.timing(4)
                    x.env.counters(reset=True)
                    torch.cuda.synchronize()
                    x0 = time.perf_counter()
                    for _ in range(ksteps):
                        x.step()
                    torch.cuda.synchronize()
                    xdt = time.perf_counter() - x0
                    pk = x.read_timing()
                    cn = x.env.counters(reset=True)
                    x.env.check_errors()
                    extra[name] = {"value": x.n * ksteps / xdt, "unit": "env-steps/s", "envs": x.n, "steps": ksteps, "warmup": kwarm, "preroll": kpre, "ms_per_step": xdt / ksteps * 1e3,
                                   "workload": "%s, %s-image obs [N,%d,%d,%d] f32" % (x.desc, x.obs_kind, x.env.channels, x.env.height, x.env.width),
                                   "algo_bytes_per_env_step": x.algo_bytes, "achieved_end_to_end_GBps": x.algo_bytes * x.n / (xdt / ksteps) / 1e9,
                                   "frac_end_to_end": x.algo_bytes * x.n / (xdt / ksteps) / 1e9 / HBM_PEAK_GBPS, "per_kernel": pk,
                                   "rates_per_s": {k: v / xdt for k, v in cn.items()}}
                    if name == "default":
                        extra[name]["step_us"] = x.step_percentiles(500)
                    x.close()
                    del x
                    if name == "default" and not args.no_cpu_baseline:  # the CPU number beside config 3 too (bounded: ~6 s)
                        cores = os.cpu_count() or 1
                        cv, cs, cdt, cu = cpu_sample(golden_config("default"), 32768 if cores >= 64 else 4096, cores, 6.0)
                        extra[name]["cpu_baseline"] = {"value": cv, "unit": "env-steps/s", "cores": cu, "kind": "port",
                                                       "sample": "%d envs x %d lock-step steps of config-default.json 80x24, C oracle, %d pthreads, %.1f s" % (32768 if cores >= 64 else 4096, cs, cu, cdt)}
                except Exception as e:
                    extra[name] = {"error": str(e)}
            out["extra_workloads"] = extra
            # The same workload with ROGUE_GYM_HIP_KEEP_SPARES=1 (opt-in, off by default and off for `value`): GameConfig::build is a pure function of
            # config and seed, so an env with a FIXED seed resets to the same level-1 state every episode and its pre-built spare need not be consumed and
            # generated again.  `value` regenerates every consumed spare, as the reference rebuilds every RunTime; this is the rate without that work.
            try:
                os.environ["ROGUE_GYM_HIP_KEEP_SPARES"] = "1"
                try:
                    x = Harness(torch, "mini", 0, 0, local_rank)
                finally:
                    del os.environ["ROGUE_GYM_HIP_KEEP_SPARES"]
                for _ in range(1500 + 50):
                    x.step()
                x.timing(7)
                x.env.counters(reset=True)
                torch.cuda.synchronize()
                x0 = time.perf_counter()
                for _ in range(600):
                    x.step()
                torch.cuda.synchronize()
                xdt = time.perf_counter() - x0
                pk = x.read_timing()
                cn = x.env.counters(reset=True)
                x.env.check_errors()
                out["fixed_seed_spares_kept"] = {"value": x.n * 600 / xdt, "unit": "env-steps/s", "steps": 600, "preroll": 1500, "warmup": 50, "ms_per_step": xdt / 600 * 1e3,
                                                 "per_kernel": pk, "rates_per_s": {k: v / xdt for k, v in cn.items()},
                                                 "note": "opt-in ROGUE_GYM_HIP_KEEP_SPARES=1: resets of fixed-seed envs copy an immutable pre-built level-1 state (bit-identical "
                                                         "results, tests/test_gpu_features.py::test_kept_spares_*); NOT the headline, which regenerates every consumed spare"}
                x.close()
                del x
            except Exception as e:
                out["fixed_seed_spares_kept"] = {"error": str(e)}
        if args.workload == "mini" and not args.no_extra:
            # Opt-in, NOT the headline: the observation tensor BOUND to the stepper (rg_obs_bind; HipVecRogueEnv(persistent_obs=True)) -- the same f32 batch in the same
            # buffer, kept current in place: the observation pass rewrites only the envs whose screen changed in this step.  `value` re-encodes every env every step.
            try:
                x = Harness(torch, "mini", 0, 0, local_rank, persistent_obs=True)
                for _ in range(1500 + 50):
                    x.step()
                x.timing(7)
                x.env.counters(reset=True)
                torch.cuda.synchronize()
                x0 = time.perf_counter()
                for _ in range(600):
                    x.step()
                torch.cuda.synchronize()
                xdt = time.perf_counter() - x0
                pk = x.read_timing()
                x.env.check_errors()
                out["bound_observation_tensor"] = {"value": x.n * 600 / xdt, "unit": "env-steps/s", "steps": 600, "preroll": 1500, "warmup": 50, "ms_per_step": xdt / 600 * 1e3,
                                                   "per_kernel": pk,
                                                   "note": "opt-in rg_obs_bind: the caller's observation tensor is updated IN PLACE -- only the envs whose screen changed are re-encoded "
                                                           "(bit-identical contents, tests/test_gpu_features.py::test_bound_observation_tensor_is_the_full_encode); NOT the headline, "
                                                           "which writes all 65 536 images every step"}
                x.close()
                del x
            except Exception as e:
                out["bound_observation_tensor"] = {"error": str(e)}
        if not args.no_cpu_baseline and WORKLOADS[args.workload][2] == "gray":
            out["cpu_baseline"] = cpu_baseline(golden_config(WORKLOADS[args.workload][0]), WORKLOADS[args.workload][5])
    if rank == 0:
        emit_line()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
