"""ParallelRogueEnv (python/rogue_gym/envs/parallel.py:19-77) over the batched HIP stepper."""
import json
from typing import Dict, Iterable, List, Tuple, Union

from rogue_gym_python._rogue_gym import ParallelGameState, PlayerState

from ._gym_compat import Discrete
from .rogue_env import ImageSetting, RogueEnv


class ParallelRogueEnv:
    """Steps many rogue-gym environments in lock-step; terminal envs restart automatically."""

    metadata = RogueEnv.metadata
    SYMBOLS = RogueEnv.SYMBOLS
    ACTION_MEANINGS = RogueEnv.ACTION_MEANINGS
    ACTIONS = RogueEnv.ACTIONS
    ACTION_LEN = len(ACTIONS)

    def __init__(self, config_dicts: Iterable[dict], max_steps: int = 1000, image_setting: ImageSetting = ImageSetting()) -> None:
        config_dicts = list(config_dicts)
        self.game = ParallelGameState(max_steps, [json.dumps(d) for d in config_dicts])
        self.result = None
        self.max_steps = max_steps
        self.steps = 0
        self.action_space = Discrete(self.ACTION_LEN)
        self.observation_space = image_setting.detect_space(*self.game.screen_size(), self.game.symbols())
        self.image_setting = image_setting
        self.states = self.game.states()
        self.num_workers = len(config_dicts)

    def get_key_to_action(self) -> Dict[str, str]:
        return self.ACTION_MEANINGS

    def step(self, action: Union[Iterable[int], str]) -> Tuple[List[PlayerState], List[float], List[bool], List[dict]]:
        """`action`: a string of num_workers keys, or an iterable of action indices."""
        if isinstance(action, str) and len(action) == self.num_workers:
            action = [ord(c) for c in action]
        else:
            try:
                action = [ord(self.ACTIONS[x]) for x in action]
            except Exception:
                raise ValueError("Invalid action: {}".format(action))
        states = self.game.step(action)
        rewards = [max(0, after.gold - before.gold) for before, after in zip(self.states, states)]
        done = [s.is_terminal for s in states]
        self.states = states
        return self.states, rewards, done, [{}] * self.num_workers

    def reset(self) -> List[PlayerState]:
        self.states = self.game.reset()
        return self.states

    def close(self) -> None:
        self.game.close()

    def seed(self, seeds: List[int]) -> None:
        self.game.seed(seeds)
