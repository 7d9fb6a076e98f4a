"""Per-kernel timing experiments on the HIP stepper (development aid; uses the C-ABI timing API)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rogue-gym_amd"))
import numpy as np
import torch
from rogue_gym_python import _rogue_gym as inner

G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))


def run(name, cfg, keys_table, n=65536, steps=300, max_steps=1000, resets=0):
    cfgs = [json.dumps(dict(cfg, seed=i)) for i in range(n)]
    h = inner._Handle(cfgs, max_steps, True)
    L = h.L
    dev = torch.device("cuda", 0)
    table = torch.tensor(list(keys_table), dtype=torch.uint8, device=dev)
    obs = torch.empty((n, 1, h.height, h.width), dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    keys = table[torch.randint(0, len(keys_table), (64, n), generator=gen, device=dev)].contiguous()
    for t in range(50):
        L.rg_step(h.h, C.c_void_p(keys[t % 64].data_ptr()), 1)
        L.rg_obs_gray(h.h, 0, 0, C.c_void_p(obs.data_ptr()))
    L.rg_timing_enable(h.h, 1)
    for t in range(steps):
        L.rg_step(h.h, C.c_void_p(keys[t % 64].data_ptr()), 1)
        L.rg_obs_gray(h.h, 0, 0, C.c_void_p(obs.data_ptr()))
    for _ in range(resets):
        L.rg_reset(h.h)
    ms = (C.c_double * 4)()
    cnt = (C.c_uint64 * 4)()
    L.rg_timing_read(h.h, ms, cnt)
    out = {k: round(ms[i] / max(cnt[i], 1) * 1e3, 1) for i, k in enumerate(["step", "render", "gray", "build"])}
    print("%-28s n=%d %s us" % (name, n, out), flush=True)
    h.close()


if __name__ == "__main__":
    A11 = b".hjklnbuy>s"
    which = sys.argv[1:] or ["all"]
    if "all" in which or "base" in which:
        run("mini 11-act", G["configs"]["mini"], A11, resets=3)
        run("mini noop only", G["configs"]["mini"], b".")
        run("mini moves only", G["configs"]["mini"], b"hjklyubn")
        run("mini 11-act max_steps=1e6", G["configs"]["mini"], A11, max_steps=1000000)
        run("mini no-enemy 11-act", dict(G["configs"]["mini"], enemies={"enemies": []}), A11, resets=3)
        run("mini no-enemy max_steps=1e6", dict(G["configs"]["mini"], enemies={"enemies": []}), A11, max_steps=1000000)
    if "all" in which or "default" in which:
        run("default 11-act n=32768", G["configs"]["default"], A11, n=32768, resets=2)


def prof(name, cfg, keys_table, n=65536, steps=200, max_steps=1000, do_reset=False):
    cfgs = [json.dumps(dict(cfg, seed=i)) for i in range(n)]
    h = inner._Handle(cfgs, max_steps, True)
    L = h.L
    L.rg_prof.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    dev = torch.device("cuda", 0)
    table = torch.tensor(list(keys_table), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    keys = table[torch.randint(0, len(keys_table), (64, n), generator=gen, device=dev)].contiguous()
    for t in range(100):
        L.rg_step(h.h, C.c_void_p(keys[t % 64].data_ptr()), 1)
    L.rg_prof(h.h, 1, None)
    if do_reset:
        L.rg_reset(h.h); steps = 1
    else:
        for t in range(steps):
            L.rg_step(h.h, C.c_void_p(keys[t % 64].data_ptr()), 1)
    out = (C.c_ulonglong * 128)()
    L.rg_prof(h.h, 0, out)
    nw = (n + 63) // 64
    names = {0: "load", 1: "pre-gen", 2: "gen_service", 3: "player+prepass", 4: "bfs", 5: "monsters", 6: "tail", 8: "g.clear", 9: "g.rooms", 10: "g.paint",
             11: "g.passages", 12: "g.corridors", 13: "g.gold", 14: "g.stair", 15: "g.monsters", 16: "g.place+rest", 17: "g.copyout"}
    print(name)
    for i in sorted(names):
        if out[32 + i]:
            print("   %-16s max %8.1f us (over all launches)   avg/wave/launch %8.2f us" % (names[i], out[i] / 100.0, out[32 + i] / 100.0 / nw / steps))
    ngen = out[32 + 21] + out[32 + 23]
    if ngen:
        print("   per-generation averages (us, ticks/2400): " + "  ".join("%s %.1f" % (names[i][2:], out[32 + i] / ngen / 2400.0) for i in range(8, 16))
              + "  | place+copy %.1f" % ((out[32 + 20] + out[32 + 22]) / ngen / 2400.0 - sum(out[32 + i] for i in range(9, 16)) / ngen / 2400.0))
    if not do_reset:
        print("   wave-duration histogram (16 us buckets, waves per launch): " + " ".join("%.0f" % (out[64 + b] / steps) for b in range(16)))
    for nm, k in (("gen (1 lane)", 20), ("gen (>1 lanes)", 22), ("bfs", 24)):
        if out[32 + k + 1]:
            print("   %-16s count/launch %.1f  avg %.1f ticks  max %.1f ticks" % (nm, out[32 + k + 1] / steps, out[32 + k] / out[32 + k + 1], out[k]))
    h.close()


if __name__ == "__main__" and "prof" in sys.argv[1:]:
    prof("k_step mini 11-act", G["configs"]["mini"], b".hjklnbuy>s")
    prof("k_build mini", G["configs"]["mini"], b".", do_reset=True)


def single_gen(name, cfg, n=1, reps=200):
    cfgs = [json.dumps(dict(cfg, seed=i)) for i in range(n)]
    h = inner._Handle(cfgs, 1000, True)
    L = h.L
    for _ in range(20):
        L.rg_reset(h.h)
    L.rg_timing_enable(h.h, 1)
    for _ in range(reps):
        L.rg_reset(h.h)
    ms = (C.c_double * 4)()
    cnt = (C.c_uint64 * 4)()
    L.rg_timing_read(h.h, ms, cnt)
    print("%-20s n=%d k_build avg %.1f us" % (name, n, ms[3] / cnt[3] * 1e3), flush=True)
    h.close()


if __name__ == "__main__" and "gen1" in sys.argv[1:]:
    for n in (1, 2, 8, 64, 64 * 256):
        single_gen("mini", G["configs"]["mini"], n)
    single_gen("default", G["configs"]["default"], 1)
    single_gen("mini-noenemy", dict(G["configs"]["mini"], enemies={"enemies": []}), 1)


def obs_only(name, cfg, n=65536, reps=300, step_every=0):
    """k_obs in isolation: without steps in between no env redraws (mirror path: 512 B read + 2 KB write per mini env)."""
    cfgs = [json.dumps(dict(cfg, seed=i)) for i in range(n)]
    h = inner._Handle(cfgs, 1000, True)
    L = h.L
    dev = torch.device("cuda", 0)
    obs = torch.empty((n, 1, h.height, h.width), dtype=torch.float32, device=dev)
    table = torch.tensor(list(b".hjklnbuy>s"), dtype=torch.uint8, device=dev)
    keys = table[torch.randint(0, 11, (64, n), device=dev)].contiguous()
    for t in range(100):
        L.rg_step(h.h, C.c_void_p(keys[t % 64].data_ptr()), 1)
        L.rg_obs_gray(h.h, 0, 0, C.c_void_p(obs.data_ptr()))
    torch.cuda.synchronize()
    L.rg_timing_enable(h.h, 1)
    for t in range(reps):
        if step_every and t % step_every == 0:
            L.rg_step(h.h, C.c_void_p(keys[t % 64].data_ptr()), 1)
        L.rg_obs_gray(h.h, 0, 0, C.c_void_p(obs.data_ptr()))
    ms = (C.c_double * 4)()
    cnt = (C.c_uint64 * 4)()
    L.rg_timing_read(h.h, ms, cnt)
    print("%-34s n=%d k_obs avg %.1f us (step avg %.1f us)" % (name, n, ms[2] / cnt[2] * 1e3, ms[0] / max(cnt[0], 1) * 1e3), flush=True)
    h.close()


if __name__ == "__main__" and "obs" in sys.argv[1:]:
    obs_only("mini obs-only (no redraw)", G["configs"]["mini"])
    obs_only("mini step+obs", G["configs"]["mini"], step_every=1)
    obs_only("mini step + 2 obs", G["configs"]["mini"], step_every=2)
