"""MI355X-native drop-in for the reference `rogue_gym` package (python/rogue_gym/__init__.py)."""
from .envs import (DungeonType, FirstFloorEnv, HipVecRogueEnv, HipVecStairReward, ImageSetting, ParallelRogueEnv, PlayerState, RogueEnv, StairRewardEnv, StairRewardParallel,
                   StatusFlag)
from . import envs  # noqa: F401

__version__ = "0.0.2"
