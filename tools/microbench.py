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
    if "nodesc" in which:
        run("mini 11-act", G["configs"]["mini"], A11)
        run("mini 10-act (no >)", G["configs"]["mini"], b".hjklnbuys")
        run("mini 10-act (no >) no-enemy", dict(G["configs"]["mini"], enemies={"enemies": []}), b".hjklnbuys")
        run("mini 10-act (no >) max_steps=1e6", G["configs"]["mini"], b".hjklnbuys", max_steps=1000000)
    if "all" in which or "default" in which:
        run("default 11-act n=32768", G["configs"]["default"], A11, n=32768, resets=2)


PHASES = {0: "load", 26: "window load", 1: "pre-gen/post-loop", 2: "gen_service", 28: "player action", 29: "turn_passed", 30: "mon prepass", 3: "dist lookup",
          27: "fill+flush", 4: "bfs", 5: "monsters", 6: "tail", 7: "stores + spare take"}
GEN_PHASES = {8: "g.clear", 9: "g.rooms", 10: "g.paint", 11: "g.passages", 12: "g.corridors", 13: "g.gold", 14: "g.stair", 15: "g.monsters", 17: "g.reveal", 18: "g.place",
              19: "g.hand-back", 21: "g.barrier", 16: "g.copy-out", 23: "g.barrier2", 20: "g.total",
              # -DRG_FINE_PROF builds only: inside connect_rooms (40 = everything between two connects) and the monster loop
              40: "f.tree/between", 41: "f.door1", 42: "f.door2", 43: "f.bend+record", 44: "f.mon.loop", 45: "f.mon.select", 46: "f.mon.appear+type", 47: "f.mon.hp",
              48: "f.mon.store"}
TICK_US = 1.0 / 2350.0  # s_memtime ticks at the shader clock (~2.35 GHz under this load; calibrated against the HIP-event kernel duration)


def prof(name, cfg, keys_table, n=65536, launches=20, max_steps=1000, do_reset=False, warm=150):
    """Per-wave phase trace of k_step (or k_build with do_reset): rows of (phase, ticks) records, aggregated here."""
    cfgs = [json.dumps(dict(cfg, seed=i)) for i in range(n)]
    h = inner._Handle(cfgs, max_steps, True)
    L = h.L
    L.rg_prof.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    dev = torch.device("cuda", 0)
    table = torch.tensor(list(keys_table), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    keys = table[torch.randint(0, len(keys_table), (64, n), generator=gen, device=dev)].contiguous()
    bound = None
    if os.environ.get("RG_PROF_BOUND"):  # the step-kernel instance of a handle with a bound observation tensor (rg_obs_bind): its tensor is kept current after every step
        bound = torch.empty((n, 1, h.height, h.width), dtype=torch.float32, device=dev)
        h.check(L.rg_obs_bind(h.h, 0, 0, 0, C.c_void_p(bound.data_ptr())))
        h.check(L.rg_obs_gray(h.h, 0, 0, C.c_void_p(bound.data_ptr())))
    for t in range(warm):
        L.rg_step(h.h, C.c_void_p(keys[t % 64].data_ptr()), 1)
        if bound is not None:
            L.rg_obs_gray(h.h, 0, 0, C.c_void_p(bound.data_ptr()))
    nw = (n + 15) // 16  # rows: one per wave of the launch (16..64 envs per wave); unused rows stay zero
    buf = np.zeros(((n + 15) // 16, 64), np.uint64)  # rg_prof copies one row per 16 envs (the smallest envs-per-wave)
    L.rg_prof(h.h, 1, None)
    sums, maxs, totals, counts = {}, {}, [], {}
    n_waves = 0
    L.rg_timing_enable(h.h, 1)
    for t in range(1 if do_reset else launches):
        if do_reset:
            L.rg_reset(h.h)
        else:
            L.rg_step(h.h, C.c_void_p(keys[t % 64].data_ptr()), 1)
            if bound is not None:
                L.rg_obs_gray(h.h, 0, 0, C.c_void_p(bound.data_ptr()))
        L.rg_prof(h.h, 1, buf.ctypes.data_as(C.c_void_p))
        buf_used = buf[buf[:, 63] > 0]
        n_waves = max(n_waves, len(buf_used))
        k = buf_used[:, 0].astype(np.int64)
        totals.append(buf_used[:, 63].astype(np.float64))
        rec = buf_used[:, 1:62]
        valid = np.arange(61)[None, :] < k[:, None]
        ph = (rec >> np.uint64(48)).astype(np.int64)
        val = (rec & np.uint64((1 << 48) - 1)).astype(np.float64)
        for pid in np.unique(ph[valid]):
            sel = valid & (ph == pid)
            per_wave = (val * sel).sum(axis=1)
            sums[pid] = sums.get(pid, 0.0) + per_wave.sum()
            maxs[pid] = max(maxs.get(pid, 0.0), per_wave.max())
            counts[pid] = counts.get(pid, 0) + int(sel.sum())
    ms = (C.c_double * 4)()
    cnt = (C.c_uint64 * 4)()
    L.rg_timing_read(h.h, ms, cnt)
    nl = len(totals)
    tot = np.concatenate(totals)
    kidx = 3 if do_reset else 0
    print("%s: kernel (HIP events) %.1f us; slowest wave avg %.1f us, mean wave %.1f us" %
          (name, ms[kidx] / max(cnt[kidx], 1) * 1e3, np.mean([t.max() for t in totals]) * TICK_US, tot.mean() * TICK_US))
    names = GEN_PHASES if do_reset else PHASES
    for pid, nm in names.items():
        if pid in sums:
            print("   %-18s avg/wave %7.2f us   max wave %7.2f us" % (nm, sums[pid] / max(n_waves, 1) / nl * TICK_US, maxs[pid] * TICK_US))
    if not do_reset:
        last = buf[:, 63].astype(np.float64) * TICK_US
        for nm, lo, hi in (("stair blocks", 0, 256), ("index-order blocks", 256, len(last))):
            seg = last[lo:hi]; seg = seg[seg > 0]
            if len(seg):
                print("   %-12s (last launch) n=%d mean %.1f us p50 %.1f p90 %.1f max %.1f" % (nm, len(seg), seg.mean(), np.percentile(seg, 50), np.percentile(seg, 90), seg.max()))
        hist = np.histogram(tot * TICK_US, bins=np.arange(0, 260, 10))[0] / nl
        print("   wave-duration histogram (10 us buckets, waves per launch): " + " ".join("%.0f" % v for v in hist))
        # where the launch's time goes outside its waves: start / end of every wave of the last launch on the chip-wide 100 MHz clock
        used = buf[:, 63] > 0
        st = (buf[used, 62] >> np.uint64(32)).astype(np.int64); en = (buf[used, 62] & np.uint64(0xffffffff)).astype(np.int64)
        en = en + ((en < st) * (1 << 32))
        idx = np.nonzero(used)[0]
        t0 = st.min()
        s_us = (st - t0) / 100.0; e_us = (en - t0) / 100.0
        print("   wave starts after the first wave's start (us): p50 %.1f p90 %.1f p99 %.1f max %.1f; last wave ends at %.1f us" %
              (np.percentile(s_us, 50), np.percentile(s_us, 90), np.percentile(s_us, 99), s_us.max(), e_us.max()))
        for nm, lo, hi in (("stair", 0, 256), ("index 0-767", 256, 1024), ("index 768+", 1024, 1 << 30)):
            sel = (idx >= lo) & (idx < hi)
            if sel.any():
                print("      %-12s start p50 %.1f max %.1f | end p50 %.1f p99 %.1f max %.1f" % (nm, np.percentile(s_us[sel], 50), s_us[sel].max(), np.percentile(e_us[sel], 50), np.percentile(e_us[sel], 99), e_us[sel].max()))
        late = np.argsort(-e_us)[:5]
        print("      last to end: " + ", ".join("block %d start %.1f end %.1f" % (idx[i], s_us[i], e_us[i]) for i in late))
        # which phase makes the slowest waves slow: phase sums over the slowest 1 % of waves of the last launch
        for nm, pid in (("gen (1 lane)", 20), ("gen (>1 lanes)", 22), ("bfs service", 24)):
            if pid in sums:
                print("   %-16s calls/launch %.1f  avg %.1f us" % (nm, counts[pid] / nl, sums[pid] / counts[pid] * TICK_US))
        if 25 in sums:
            print("   dist maps/launch %.1f" % (sums[25] / nl))
        order = np.argsort(-buf[:, 63].astype(np.float64))[:int(os.environ.get('RG_SLOW_WAVES', '3'))]  # rows of the last launch
        for wv in order:  # the slowest waves of the last launch, record by record
            kk = int(buf[wv, 0])
            recs = ["%s=%.1f" % (PHASES.get(int(r >> np.uint64(48)), GEN_PHASES.get(int(r >> np.uint64(48)), str(int(r >> np.uint64(48))))), float(r & np.uint64((1 << 48) - 1)) * TICK_US) for r in buf[wv, 1:1 + kk]]
            print("   slow wave %d: total %.1f us: %s" % (wv, float(buf[wv, 63]) * TICK_US, " ".join(recs)))
    h.close()


if __name__ == "__main__" and "profd" in sys.argv[1:]:
    prof("k_step default 80x24 11-act, 32768 envs", G["configs"]["default"], b".hjklnbuy>s", n=32768)


if __name__ == "__main__" and "prof1" in sys.argv[1:]:
    prof("k_step mini 11-act after 1500 steps", G["configs"]["mini"], b".hjklnbuy>s", warm=1500)


if __name__ == "__main__" and "profphase" in sys.argv[1:]:
    for warm in (5, 150, 1500, 3000):
        prof("k_step mini 11-act after %d steps" % warm, G["configs"]["mini"], b".hjklnbuy>s", warm=warm)


if __name__ == "__main__" and "prof" in sys.argv[1:]:
    prof("k_step mini 11-act", G["configs"]["mini"], b".hjklnbuy>s")
    prof("k_step mini 10-act (no >)", G["configs"]["mini"], b".hjklnbuys")
    prof("k_step mini 10-act (no >) no-enemy", dict(G["configs"]["mini"], enemies={"enemies": []}), b".hjklnbuys")
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


if __name__ == "__main__" and "profb" in sys.argv[1:]:
    prof("k_build mini (16 envs per wave)", G["configs"]["mini"], b".", n=4096, do_reset=True)


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
