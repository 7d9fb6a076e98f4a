from .parallel import ParallelRogueEnv
from .rogue_env import DungeonType, ImageSetting, PlayerState, RogueEnv, StatusFlag
from .wrappers import FirstFloorEnv, StairRewardEnv, StairRewardParallel
from .device import HipVecRogueEnv, HipVecStairReward

__all__ = ["ParallelRogueEnv", "DungeonType", "ImageSetting", "PlayerState", "RogueEnv", "StatusFlag", "FirstFloorEnv", "StairRewardEnv",
           "StairRewardParallel", "HipVecRogueEnv", "HipVecStairReward"]
