"""Transcribe the reference's golden DATA into tests/golden/*.json (run in the build container only).

Inputs are data, not code: the expected screens / key strings held by the reference's own tests
(/root/reference/python/tests/data.py), the assertion constants of test_ff_env.py / test_st_env.py /
test_parallel.py / core/src/dungeon/rogue/mod.rs:566-578, and the JSON config + replay assets under
/root/reference/data.  /root/reference does not exist on the GPU box; the committed JSON does.
"""
import hashlib
import importlib.util
import json
import os

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location("refdata", os.path.join(REF, "python/tests/data.py"))
refdata = importlib.util.module_from_spec(spec)
spec.loader.exec_module(refdata)


def load(p):
    with open(os.path.join(REF, p)) as f:
        return json.load(f)


ACT_TO_KEY = {"Right": "l", "Up": "k", "Down": "j", "Left": "h", "RightUp": "u", "LeftUp": "y", "RightDown": "n", "LeftDown": "b"}


def actions_to_keys(acts):
    """data/learned/.../best-actions.json ([{"Act": {"Move": "Right"}}, ...]) -> ai-keymap string (input.rs:73-100)."""
    out = []
    for a in acts:
        act = a["Act"]
        if act == "Search":
            out.append("s")
        elif act == "DownStair":
            out.append(">")
        elif act == "NoOp":
            out.append(".")
        elif "Move" in act:
            out.append(ACT_TO_KEY[act["Move"]])
        elif "MoveUntil" in act:
            out.append(ACT_TO_KEY[act["MoveUntil"]].upper())
        else:
            raise ValueError(act)
    return "".join(out)


golden = {
    "_source": "python/tests/data.py, test_ff_env.py:14-22, test_st_env.py:27-37, test_parallel.py:51-60, core/src/dungeon/rogue/mod.rs:566-578, core/src/dungeon/rogue/passages.rs:272-296",
    "screens": {
        "SEED1_DUNGEON2": refdata.SEED1_DUNGEON2,
        "SEED1_DUNGEON3": refdata.SEED1_DUNGEON3,
        "SEED1_DUNGEON_CLEAR": refdata.SEED1_DUNGEON_CLEAR,
        "SEED1_DUNGEON_STALE": refdata.SEED1_DUNGEON,  # 21 rows: stale at this revision
    },
    "keys": {
        "CMD_STR": refdata.CMD_STR, "CMD_STR2": refdata.CMD_STR2, "CMD_STR3": refdata.CMD_STR3,
        "CMD_STR4": refdata.CMD_STR4, "CMD_STR5": refdata.CMD_STR5,
    },
    "configs": {
        "seed1": {"seed": 1},
        "seed1_noenem": {"seed": 1, "enemies": {"enemies": []}},
        "ff": {"seed": 1, "hide_dungeon": False, "enemies": {"enemies": []}},
        "st": {"width": 32, "height": 16, "seed": 5, "hide_dungeon": False,
               "dungeon": {"style": "rogue", "room_num_x": 2, "room_num_y": 2}, "enemies": {"enemies": []}},
        "move_enemy_kat": {"width": 32, "height": 16, "seed": 5,
                           "dungeon": {"style": "rogue", "room_num_x": 2, "room_num_y": 2, "min_room_size": {"x": 4, "y": 4}}},
        "mini": load("data/config-mini.json"),
        "nohide": load("data/config-nohide.json"),
        "default": load("data/config-default.json"),
        "ddqn": load("data/learned/ddqn-minidungeon/config.json"),
    },
    "expect": {
        "ff": {"keys": "CMD_STR2", "reward_with_stair100": 102, "done": True, "symbol_image_shape": [18, 24, 80]},
        "st": {"first": {"keys": "CMD_STR3", "reward_with_stair100": 104.0},
               "second": {"keys": "CMD_STR4", "reward_with_stair100": 100.0, "image_shape": [21, 16, 32],
                          "plane17": 3.0, "plane18": 12.0, "status_vec_full": [3, 12, 12, 16, 16, 0, 1, 0, 0]}},
        "move_enemy_kat": {"from": [9, 9], "to": [28, 4], "next": [10, 9]},
        # core/src/dungeon/rogue/passages.rs:272-296 `test_inclusive_edges`: RectRange 5..10 x 6..9, inclusive edges per direction
        # (pins rect-iter's corner naming: "upper" = y1 - 1)
        "inclusive_edges": {"range": {"x": [5, 10], "y": [6, 9]},
                            "Down": [[x, 8] for x in range(6, 9)], "Up": [[x, 6] for x in range(6, 9)],
                            "Left": [[5, y] for y in range(7, 8)], "Right": [[9, y] for y in range(7, 8)]},
        "shapes": {"symbol_hist_noenem": [18, 24, 80], "gray": [1, 24, 80], "gray_hist": [2, 24, 80], "space_noenem_full": [26, 24, 80]},
    },
    "ddqn_keys": actions_to_keys(load("data/learned/ddqn-minidungeon/best-actions.json")),
    # digest of the reference's own serialisation of that key log (RunTime::saved_inputs_as_json = serde_json::to_string_pretty): the
    # action-history dump of the HIP stepper must reproduce the file byte for byte (trailing newline excluded)
    "ddqn_actions_sha256": hashlib.sha256(open(os.path.join(REF, "data/learned/ddqn-minidungeon/best-actions.json")).read().rstrip("\n").encode()).hexdigest(),
}

with open(os.path.join(OUT, "reference_goldens.json"), "w") as f:
    json.dump(golden, f, indent=1)
print("wrote", os.path.join(OUT, "reference_goldens.json"))
