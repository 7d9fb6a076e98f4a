import json, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rogue-gym_amd"), os.path.join(ROOT, "tests")]
from parity_util import HipBatch, make_oracles
for rx, ry in ((8, 4), (10, 4)):
    cfg = {"width": 160, "height": 48, "dungeon": {"style": "rogue", "room_num_x": rx, "room_num_y": ry}}
    seeds = list(range(1000, 1004))
    hip = HipBatch(cfg, seeds); orc = make_oracles(cfg, seeds)
    scr, hist, st, fl = hip.fetch()
    for i, o in enumerate(orc):
        d, cells = hip.debug(i)
        surf, attr, doors, gold = o.grid()
        diff = np.argwhere((cells & 7) != surf)
        rs, cnt = o.rng()
        same_rng = list(d.rng) == [int(v) for v in rs.reshape(-1)]
        print(rx, ry, "seed", seeds[i], "surface diffs", len(diff), "first", diff[:3].tolist(), "rng same", same_rng, "player", (d.px, d.py), (o.scalars()["px"], o.scalars()["py"]))
        if len(diff) and i == 0:
            rooms = [(k, d.room_rect[k] & 0xff, (d.room_rect[k] >> 8) & 0xff, (d.room_rect[k] >> 16) & 0xff, d.room_rect[k] >> 24, d.room_meta[k]) for k in range(d.n_rooms)]
            print(" hip rooms", rooms)
        mons = o.monsters()
        got_m = [(d.mon_x[k], d.mon_y[k], d.mon_type[k], d.mon_hp[k]) for k in range(d.n_monsters)]
        exp_m = [(m["x"], m["y"], m["type"], m["hp"]) for m in mons]
        if got_m != exp_m:
            print("  monsters differ: hip", len(got_m), "oracle", len(exp_m))
            print("   hip", got_m[:12]); print("   orc", exp_m[:12])
