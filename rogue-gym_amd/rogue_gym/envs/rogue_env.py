"""Single-environment gym surface of the HIP stepper: `RogueEnv`, `ImageSetting`, `StatusFlag`, `DungeonType`.

The public names, constructor arguments, return shapes and numeric conventions are those of the reference package
(/root/reference/python/rogue_gym/envs/rogue_env.py; pinned by python/tests/*.py, re-expressed in tests/test_gpu_reference_suite.py),
so code written against `rogue_gym.envs` runs unchanged.  The implementation is this repository's own: every observation is produced by
librogue_gym_hip.so on the GPU (rogue_gym_python._rogue_gym), and image requests on states that came out of one batch are served by a
single whole-batch kernel launch (StateBatch.images) instead of one encode per state.
"""
import enum
import json
from typing import Dict, List, NamedTuple, Optional, Sequence, Tuple, Union

import numpy as np

from rogue_gym_python import _rogue_gym as rogue_gym_inner
from rogue_gym_python._rogue_gym import GameState, PlayerState, StateBatch

from ._gym_compat import Box, Discrete, Env

# (kind, with_hist) -> PlayerState method; kind 0 = gray, 1 = symbol (python/src/lib.rs:162-205)
_IMAGE_METHODS = {
    (0, False): "gray_image", (0, True): "gray_image_with_hist",
    (1, False): "symbol_image", (1, True): "symbol_image_with_hist",
}


def _read_text(path: str) -> str:
    with open(path, "r") as fh:
        return fh.read()


def _write_text(path: str, text: str) -> None:
    with open(path, "w") as fh:
        fh.write(text)


def _require_state(obj) -> None:
    if not isinstance(obj, PlayerState):
        raise TypeError("Needs PlayerState, but {} was given".format(type(obj)))


class StatusFlag(enum.Flag):
    """Which entries of the player status become constant image planes / vector entries.  Bit b selects entry b of
    [dungeon_level, hp_current, hp_max, str_current, str_max, defense, player_level, exp, hunger] (StatusFlagInner, python/src/flags.rs:45-55)."""

    EMPTY = 0
    DUNGEON_LEVEL = 1 << 0
    HP_CURRENT = 1 << 1
    HP_MAX = 1 << 2
    STR_CURRENT = 1 << 3
    STR_MAX = 1 << 4
    DEFENSE = 1 << 5
    PLAYER_LEVEL = 1 << 6
    EXP = 1 << 7
    HUNGER = 1 << 8
    FULL = (1 << 9) - 1

    def count_one(self) -> int:
        return bin(self.value & 0x1FF).count("1")

    def _image_of(self, state: PlayerState, kind: int, with_hist: bool) -> np.ndarray:
        _require_state(state)
        return getattr(state, _IMAGE_METHODS[(kind, with_hist)])(flag=self.value)

    def symbol_image(self, state: PlayerState) -> np.ndarray:
        return self._image_of(state, 1, False)

    def symbol_image_with_hist(self, state: PlayerState) -> np.ndarray:
        return self._image_of(state, 1, True)

    def gray_image(self, state: PlayerState) -> np.ndarray:
        return self._image_of(state, 0, False)

    def gray_image_with_hist(self, state: PlayerState) -> np.ndarray:
        return self._image_of(state, 0, True)

    def status_vec(self, state: PlayerState) -> List[int]:
        _require_state(state)
        return state.status_vec(self.value)


class DungeonType(enum.Enum):
    GRAY = 1
    SYMBOL = 2


class ImageSetting(NamedTuple):
    """How a PlayerState becomes a [C, H, W] float image: the dungeon as one gray plane or as one-hot symbol planes, followed by one constant
    plane per selected status entry and optionally the visited-history plane."""

    dungeon: DungeonType = DungeonType.SYMBOL
    status: StatusFlag = StatusFlag.FULL
    includes_hist: bool = False

    @property
    def _kind(self) -> int:
        return 1 if self.dungeon == DungeonType.SYMBOL else 0

    def dim(self, channels: int) -> int:
        return (channels if self._kind else 1) + self.status.count_one() + int(bool(self.includes_hist))

    def detect_space(self, h: int, w: int, symbols: int) -> Box:
        return Box(low=0, high=1, shape=(self.dim(symbols), h, w), dtype=np.float32)

    def expand(self, state: PlayerState) -> np.ndarray:
        return self.status._image_of(state, self._kind, bool(self.includes_hist))

    def expand_batch(self, states: Union[StateBatch, Sequence[PlayerState]]) -> np.ndarray:
        """[n, C, H, W] for a whole batch of states: one kernel launch when `states` is the StateBatch a ParallelRogueEnv returned."""
        if isinstance(states, StateBatch):
            return states.images(self._kind, self.status.value, bool(self.includes_hist))
        return np.stack([self.expand(s) for s in states])


# Symbol order of core/src/symbol.rs:17-40 / tile.rs: 17 fixed glyphs, then the 26 monster letters
_FIXED_SYMBOLS = " @#.-%+^!?])/*:=,"
# (key, meaning) in action-index order.  The j / k labels are the reference's own (python/rogue_gym/envs/rogue_env.py:145-157); the keymap that
# actually runs is KeyMap::ai (input.rs:73-100), where j moves down and k moves up.
_ACTION_TABLE = (
    (".", "NO_OPERATION"), ("h", "MOVE_LEFT"), ("j", "MOVE_UP"), ("k", "MOVE_DOWN"), ("l", "MOVE_RIGHT"), ("n", "MOVE_RIGHTDOWN"),
    ("b", "MOVE_LEFTDOWN"), ("u", "MOVE_RIGHTUP"), ("y", "MOVE_LEFTUP"), (">", "DOWNSTAIR"), ("s", "SEARCH"),
)


class RogueEnv(Env):
    """One Rogue game (GameState: no auto-reset; after death every further action key raises)."""

    metadata = {"render.modes": ["human", "ascii"]}
    SYMBOLS = list(_FIXED_SYMBOLS) + [chr(ord("A") + i) for i in range(26)]
    ACTION_MEANINGS = dict(_ACTION_TABLE)
    ACTIONS = [key for key, _ in _ACTION_TABLE]
    ACTION_LEN = len(ACTIONS)

    def __init__(self, config_path: Optional[str] = None, config_dict: Optional[dict] = None, max_steps: int = 1000,
                 image_setting: ImageSetting = ImageSetting(), **kwargs) -> None:
        Env.__init__(self)
        if config_path is not None and config_path != "":
            config_json = _read_text(config_path)
        else:  # keyword arguments are top-level GameConfig fields (seed=..., width=..., ...) laid over config_dict
            config_json = json.dumps({**(config_dict or {}), **kwargs})
        self.game = GameState(max_steps, config_json)
        self.max_steps, self.image_setting = max_steps, image_setting
        height, width = self.game.screen_size()
        self.action_space = Discrete(self.ACTION_LEN)
        self.observation_space = image_setting.detect_space(height, width, self.game.symbols())
        self.result: PlayerState = self.game.prev()

    # ---- queries ----
    def screen_size(self) -> Tuple[int, int]:
        """(height, width)"""
        return self.game.screen_size()

    def get_key_to_action(self) -> Dict[str, str]:
        return self.ACTION_MEANINGS

    def get_dungeon(self) -> List[str]:
        """The screen as `height` strings of `width` characters."""
        return list(self.result.dungeon)

    def get_config(self) -> dict:
        return json.loads(self.game.dump_config())

    def save_config(self, fname: str) -> None:
        _write_text(fname, self.game.dump_config())

    def save_actions(self, fname: str) -> None:
        """The keys of this episode as the reference's InputCode JSON (replayable by its devui / act2gif tools)."""
        _write_text(fname, self.game.dump_history())

    def state_to_image(self, state: PlayerState, setting: Optional[ImageSetting] = None) -> np.ndarray:
        return (self.image_setting if setting is None else setting).expand(state)

    # ---- stepping ----
    def _keys_of(self, action: Union[int, str]) -> str:
        if isinstance(action, (str, bytes)):
            return action if isinstance(action, str) else action.decode("latin-1")
        try:
            return self.ACTIONS[action]
        except (IndexError, TypeError) as why:
            raise ValueError("Invalid action: {} causes {}".format(action, why)) from None

    def step(self, action: Union[int, str]) -> Tuple[PlayerState, float, bool, dict]:
        """`action` is an index into ACTIONS or a string of raw keys ("hjk", "hh>"): every key of the string is played and the reward is
        the gold collected over the whole string."""
        purse = self.result.gold
        for key in self._keys_of(action):
            self.game.react(ord(key))
        state = self.result = self.game.prev()
        return state, state.gold - purse, state.is_terminal, {}

    def seed(self, seed: int) -> None:
        """Takes effect at the next reset."""
        self.game.set_seed(seed)

    def reset(self) -> PlayerState:
        self.game.reset()
        state = self.result = self.game.prev()
        return state

    def render(self, mode: str = "human", close: bool = False) -> None:
        """Prints the screen and the status line (both modes)."""
        print(repr(self.result))

    def replay(self, interval_ms: int = 100) -> None:
        """Needs the reference's terminal UI: raises (out of scope for the HIP stepper)."""
        rogue_gym_inner.replay(self.game, interval_ms=interval_ms)

    def play_cli(self) -> None:
        """Needs the reference's terminal UI: raises (out of scope for the HIP stepper)."""
        rogue_gym_inner.play_cli(game=self.game)

    def __repr__(self) -> str:
        return repr(self.result)

    @property
    def unwrapped(self):
        return self
