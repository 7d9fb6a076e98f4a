"""`ParallelRogueEnv`: many Rogue games stepped in lock step by ONE batched kernel launch per key.

Same public surface as the reference's thread-per-env executor (/root/reference/python/rogue_gym/envs/parallel.py:19-77): `step` takes a
string of N keys or N action indices and returns (states, rewards, dones, infos); terminal envs are reset inside `step` and report the
post-reset state with `is_terminal` still true (python/src/thread_impls.rs:69-79); rewards are the clipped gold deltas.  The states come
back as a StateBatch -- a sequence of PlayerState backed by one pinned host snapshot -- and `images()` turns a whole batch into a
[N, C, H, W] array with one launch, so nothing in the per-step path is O(N) Python.
"""
import json
from typing import Dict, Iterable, List, Tuple, Union

import numpy as np

from rogue_gym_python._rogue_gym import ParallelGameState, StateBatch

from ._gym_compat import Discrete
from .rogue_env import ImageSetting, RogueEnv

_KEY_OF_ACTION = np.frombuffer("".join(RogueEnv.ACTIONS).encode(), np.uint8)


class ParallelRogueEnv:
    metadata = RogueEnv.metadata
    SYMBOLS = RogueEnv.SYMBOLS
    ACTION_MEANINGS = RogueEnv.ACTION_MEANINGS
    ACTIONS = RogueEnv.ACTIONS
    ACTION_LEN = RogueEnv.ACTION_LEN

    def __init__(self, config_dicts: Iterable[dict], max_steps: int = 1000, image_setting: ImageSetting = ImageSetting()) -> None:
        configs = [json.dumps(d) for d in config_dicts]
        self.num_workers, self.max_steps, self.image_setting = len(configs), max_steps, image_setting
        self.game = ParallelGameState(max_steps, configs)
        height, width = self.game.screen_size()
        self.action_space = Discrete(self.ACTION_LEN)
        self.observation_space = image_setting.detect_space(height, width, self.game.symbols())
        self.states: StateBatch = self.game.states()
        self._infos = [{}] * self.num_workers

    def get_key_to_action(self) -> Dict[str, str]:
        return self.ACTION_MEANINGS

    def get_configs(self) -> dict:
        return json.loads(self.game.dump_config())

    def _keys_of(self, action: Union[Iterable[int], str]) -> np.ndarray:
        """A string with one raw key per env, or one action index per env."""
        if isinstance(action, str) and len(action) == self.num_workers:
            return np.frombuffer(action.encode("latin-1"), np.uint8)
        idx = None
        try:
            idx = np.asarray(action if isinstance(action, np.ndarray) else list(action))
        except TypeError:
            pass
        if idx is None or idx.dtype.kind not in "iu" or (idx.size and (idx.max() >= self.ACTION_LEN or idx.min() < -self.ACTION_LEN)):
            raise ValueError("Invalid action: {}".format(action))
        return _KEY_OF_ACTION[idx]  # negative indices count from the end, like the list lookup they replace

    def step(self, action: Union[Iterable[int], str]) -> Tuple[StateBatch, List[float], List[bool], List[dict]]:
        before = self.states
        after = self.game.step(self._keys_of(action))
        m = min(len(before), len(after))
        gained = np.maximum(after.gold[:m].astype(np.int64) - before.gold[:m].astype(np.int64), 0)
        self.states = after
        return after, gained.tolist(), after.is_terminal.tolist(), self._infos

    def images(self, states: StateBatch = None) -> np.ndarray:
        """[N, C, H, W] float32 image of every env under `image_setting` (default: the states of the last step)."""
        return self.image_setting.expand_batch(self.states if states is None else states)

    def reset(self) -> StateBatch:
        batch = self.states = self.game.reset()
        return batch

    def seed(self, seeds: List[int]) -> None:
        """One seed per env, used from the next reset on."""
        self.game.seed(seeds)

    def close(self) -> None:
        self.game.close()
