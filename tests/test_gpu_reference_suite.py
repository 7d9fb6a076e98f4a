"""The reference's own Python integration tests (python/tests/*.py), re-expressed against the drop-in
`rogue_gym` package backed by the HIP stepper.  Expected values come from tests/golden/
reference_goldens.json (transcribed from python/tests/data.py); the stale 21-row SEED1_DUNGEON is not used."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _envs():
    from rogue_gym import envs
    return envs


def test_screen_size_and_kwargs():
    e = _envs()
    env = e.RogueEnv(config_dict={}, seed=1)
    assert env.screen_size() == (24, 80)
    assert len(env.get_dungeon()) == 24 and all(len(r) == 80 for r in env.get_dungeon())
    env = e.RogueEnv(config_dict={}, seed=1, width=48, height=24)  # test_kwargs_setting
    assert env.screen_size() == (24, 48)


def test_action(goldens):
    env = _envs().RogueEnv(config_dict={}, seed=1)
    res, *_ = env.step(goldens["keys"]["CMD_STR"])
    assert res.dungeon == goldens["screens"]["SEED1_DUNGEON2"]


def test_noaction():
    env = _envs().RogueEnv(config_dict={}, seed=1)
    state = env.result
    res, *_ = env.step(".")
    assert res.dungeon == state.dungeon
    assert res.status == state.status


def test_max_steps(goldens):
    env = _envs().RogueEnv(config_dict={}, seed=1, max_steps=5)
    _, _, done, _ = env.step(goldens["keys"]["CMD_STR"])
    assert done


def test_images(goldens):
    e = _envs()
    env = e.RogueEnv(config_dict=dict(goldens["configs"]["seed1_noenem"]))
    state, *_ = env.step("H")
    status = e.StatusFlag.EMPTY
    symbol_img_hist = status.symbol_image_with_hist(state)
    assert symbol_img_hist.shape == (18, 24, 80)
    assert (symbol_img_hist[-1][20][2:15] == 1.0).all()
    assert status.gray_image(state).shape == (1, 24, 80)
    assert status.gray_image_with_hist(state).shape == (2, 24, 80)


def test_space(goldens):
    from rogue_gym.envs._gym_compat import Box, Discrete
    env = _envs().RogueEnv(config_dict=dict(goldens["configs"]["seed1_noenem"]))
    assert env.action_space == Discrete(env.ACTION_LEN)
    assert env.observation_space == Box(low=0, high=1, shape=(26, 24, 80), dtype=np.float32)  # 17 symbols + 9 status


def test_first_floor_env(goldens):
    e = _envs()
    cfg = dict(goldens["configs"]["ff"])
    env = e.FirstFloorEnv(e.RogueEnv(config_dict=dict(cfg), image_setting=e.ImageSetting(status=e.StatusFlag.DUNGEON_LEVEL)), 100.0)
    assert len(env.unwrapped.get_dungeon()) == len(goldens["screens"]["SEED1_DUNGEON_CLEAR"])
    assert env.unwrapped.get_dungeon() == goldens["screens"]["SEED1_DUNGEON_CLEAR"]
    state, rewards, done, _ = env.step(goldens["keys"]["CMD_STR2"])
    assert done
    assert rewards == 102
    assert env.unwrapped.state_to_image(state).shape == (18, 24, 80)
    assert env.unwrapped.get_config() == cfg


def test_stair_reward_env(goldens):
    e = _envs()
    expand = e.ImageSetting(e.DungeonType.SYMBOL, e.StatusFlag.DUNGEON_LEVEL | e.StatusFlag.HP_CURRENT | e.StatusFlag.EXP, True)
    env = e.StairRewardEnv(e.RogueEnv(config_dict=dict(goldens["configs"]["st"]), image_setting=expand), 100.0)
    state, rewards, done, _ = env.step(goldens["keys"]["CMD_STR3"])
    assert rewards == 104.0
    state, rewards, _, _ = env.step(goldens["keys"]["CMD_STR4"])
    assert rewards == 100.0
    img = env.unwrapped.state_to_image(state)
    assert img.shape == (21, 16, 32)
    assert img[17][0][0] == 3.0
    assert img[18][0][0] == 12.0
    assert e.StatusFlag.FULL.status_vec(state) == [3, 12, 12, 16, 16, 0, 1, 0, 0]


NUM_WORKERS = 8


def test_parallel_configs(goldens):
    env = _envs().ParallelRogueEnv(config_dicts=[{"seed": 1}] * NUM_WORKERS)
    step = [goldens["keys"]["CMD_STR"], goldens["keys"]["CMD_STR5"]]
    for i in range(len(step[0])):
        env.step("".join(step[x % 2][i] for x in range(NUM_WORKERS)))
    for i, res in enumerate(env.states):
        assert res.dungeon == goldens["screens"]["SEED1_DUNGEON2" if i % 2 == 0 else "SEED1_DUNGEON3"]


def test_parallel_seed():
    env = _envs().ParallelRogueEnv(config_dicts=[{"seed": 1}] * NUM_WORKERS)
    first = [s.dungeon for s in env.states]
    assert all(d == first[0] for d in first)
    env.seed([10] * env.num_workers)
    for s in env.reset():
        assert s.dungeon != first[0]


def test_parallel_step_cyclic(goldens):
    env = _envs().ParallelRogueEnv(config_dicts=[{"seed": 1}] * NUM_WORKERS, max_steps=5)
    first = env.states[0].dungeon
    for i, c in enumerate(goldens["keys"]["CMD_STR"]):
        states, _, dones, _ = env.step(c * NUM_WORKERS)
        if i == 4:
            assert dones == [True] * NUM_WORKERS
            for res in states:
                assert res.dungeon == first  # the post-reset screen
        else:
            assert dones == [False] * NUM_WORKERS


def test_parallel_stair_reward(goldens):
    env = _envs().StairRewardParallel(config_dicts=[goldens["configs"]["st"]] * NUM_WORKERS, max_steps=30)
    for c in goldens["keys"]["CMD_STR3"]:
        _, rewards, *_ = env.step(c * NUM_WORKERS)
        assert all(r >= 0.0 for r in rewards)
    assert rewards == [50.0] * NUM_WORKERS
    for c in goldens["keys"]["CMD_STR4"]:
        _, rewards, *_ = env.step(c * NUM_WORKERS)
        assert all(r >= 0.0 for r in rewards)
    assert rewards == [50.0] * NUM_WORKERS
    for _ in range(30 - (len(goldens["keys"]["CMD_STR3"]) + len(goldens["keys"]["CMD_STR4"]))):
        _, rewards, *_ = env.step([0] * NUM_WORKERS)
        assert all(r >= 0.0 for r in rewards)


def test_dead_single_env_raises(goldens):
    """After death the engine is in the Grave modal and further action keys raise (core/src/lib.rs:301-315)."""
    e = _envs()
    rng = np.random.RandomState(0)
    for seed in range(40):
        env = e.RogueEnv(config_dict=dict(goldens["configs"]["mini"]), seed=seed, max_steps=100000)
        done = False
        for _ in range(3000):
            _, _, done, _ = env.step("hjklyubn"[rng.randint(8)])
            if done:
                break
        if done:
            with pytest.raises(RuntimeError):
                env.step("h")
            return
    pytest.fail("no death")


def test_hip_vec_env_tensor_path(goldens):
    import torch

    e = _envs()
    n = 1024
    cfgs = [dict(goldens["configs"]["mini"], seed=i) for i in range(n)]
    venv = e.HipVecRogueEnv(cfgs, max_steps=50)
    penv = e.ParallelRogueEnv(cfgs, max_steps=50)
    g = torch.Generator(device="cpu").manual_seed(0)
    for _ in range(70):
        a = torch.randint(0, 11, (n,), generator=g)
        obs, rew, done = venv.step(a.to(venv.device))
        states, rewards, dones, _ = penv.step(a.tolist())
    torch.cuda.synchronize()
    assert obs.shape == (n, 1, 16, 32) and obs.is_cuda
    assert rew.cpu().tolist() == [float(r) for r in rewards]
    assert done.cpu().tolist() == dones
    host = obs.cpu().numpy()
    for i in range(0, n, 37):
        assert np.array_equal(host[i], states[i].gray_image())
    venv.check_errors()


def test_hip_vec_stair_reward_matches_parallel_wrapper(goldens):
    """HipVecStairReward (device tensors) == StairRewardParallel (python/rogue_gym/envs/wrappers.py:45-64) on the DDQN key log, which
    takes seed 5 down the stairs at step 19."""
    import torch

    e = _envs()
    n = 64
    cfgs = [dict(goldens["configs"]["ddqn"], seed=5 if i % 4 == 0 else 100 + i) for i in range(n)]
    venv = e.HipVecStairReward(cfgs, max_steps=200, stair_reward=50.0)
    penv = e.StairRewardParallel(cfgs, max_steps=200, stair_reward=50.0)
    total = torch.zeros(n)
    total_ref = [0.0] * n
    for ch in goldens["ddqn_keys"][:120]:
        keys = torch.full((n,), ord(ch), dtype=torch.uint8, device=venv.device)
        _, rew, done = venv.step_keys(keys)
        _, rewards, dones, _ = penv.step(ch * n)
        assert rew.cpu().tolist() == [float(r) for r in rewards]
        assert done.cpu().tolist() == dones
        total += rew.cpu()
        total_ref = [a + b for a, b in zip(total_ref, rewards)]
    assert max(total_ref) >= 50.0  # the stair bonus was paid at least once
