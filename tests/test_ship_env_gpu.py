"""The reference's own test cases (tests/test_ship_env.py, 8 tests), re-expressed against the current API on the HIP
path.  The reference file is stale against its own constructors and cannot run; what it still encodes
— exact no-motion under rudder-only actions, position-before-velocity ordering, turn direction, oldest-first history,
nearest-goal reporting, exact reward values, done conditions — is asserted here with the same names.  Scenario set-up
uses the facade's `reset(spawn_point=..., goals=...)` extension (the reference tests' `reset(spawn_point=...)` /
`game.add_goal`)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STEP_PENALTY = -0.01
DEFAULT_STATE_VAL = -1
FAR = [[300, 560], [300, 565], [300, 570], [300, 575], [300, 580]]


@pytest.fixture()
def env():
    import random
    import torch
    assert torch.cuda.is_available()
    from ship_gym.ship_env import ShipEnv   # the reference's import line
    random.seed(3); np.random.seed(3)
    e = ShipEnv()
    e.reset()
    yield e
    e.close()


def test_reset(env):
    o = env.reset()
    assert env.game.player.x == env.game.bounds[0] / 2 and env.game.player.y == 25     # game.py:274
    assert np.all(o[:16] == DEFAULT_STATE_VAL) and o[16] == 300 and o[17] == 25
    assert np.all(o[22:] == DEFAULT_STATE_VAL) and o[18] == 0 and o[19] == 0
    for _ in range(3):
        env.step(env.action_space.sample())
    o = env.reset(spawn_point=(300, 212))
    assert (env.game.player.x, env.game.player.y) == (300, 212) and (o[16], o[17]) == (300, 212)
    assert env.step_count == 0 and env.cumulative_reward == 0 and len(env.game.goals) == 5


def test_done(env):
    # spawning on top of every goal: they are all consumed by the first step -> no goals left -> done
    env.reset(spawn_point=(300, 100), goals=[[310, 120]] * 5)
    o, r, done, _ = env.step(1)
    assert done and r == 1.0 and len(env.game.goals) == 0 and (o[20], o[21]) == (-1, -1)
    # one goal elsewhere: touching the others does not finish the episode
    env.reset(spawn_point=(300, 100), goals=[[310, 120]] * 4 + [[300, 400]])
    o, r, done, _ = env.step(1)
    assert not done and r == 1.0 and len(env.game.goals) == 1 and (o[20], o[21]) == (300, 400)


def test_action(env):
    start = (300.0, 40.0)
    env.reset(spawn_point=start, goals=FAR)
    player = env.game.player
    for a in (1, 2, 2, 1, 1, 2):                      # rudder-only actions: position EXACTLY unchanged
        env.step(a)
        assert (player.x, player.y) == start
    env.reset(spawn_point=start, goals=FAR)
    for i in range(10):
        env.step(0)                                    # forward
        if i > 0:
            assert player.y > start[1]                 # first thrust changes velocity only (position-first integrator)
        else:
            assert player.y == start[1]
        assert player.x == pytest.approx(start[0], abs=1e-9)
    # rudder to one side, then forward: after > 3 steps the ship has drifted to that side
    env.reset(spawn_point=start, goals=FAR)
    env.step(2)
    xs = []
    for i in range(7):
        env.step(0)
        xs.append(player.x)
        if i > 3:
            assert player.y > start[1] and player.x > start[0]
    env.reset(spawn_point=start, goals=FAR)
    env.step(1)
    for i in range(7):
        env.step(0)
        if i > 3:
            assert player.y > start[1] and player.x < start[0]
    assert env.game.player.rudder_angle == -5
    for _ in range(5):
        env.step(1)
    assert env.game.player.rudder_angle == -10         # clamp_rudder, models.py:136-140


def test_history_states(env):
    env.reset(spawn_point=(300, 84), goals=FAR)
    player = env.game.player
    last_state = None
    for _ in range(10):
        last_x, last_y = player.x, player.y
        states, _, _, _ = env.step(0)
        assert states[0] == last_x and states[1] == last_y          # oldest frame = position before the step
        assert states[16] == player.x and states[17] == player.y
        if last_state is not None:
            np.testing.assert_array_equal(states[:16], last_state[16:])
        last_state = states


def test_goal_states(env):
    goals = [[300, 100], [300, 170], [300, 240], [300, 310], [300, 380]]
    o = env.reset(spawn_point=(295, 30), goals=goals)
    assert (o[20], o[21]) == (300, 100)                             # nearest goal in the reset observation
    seen = []
    for _ in range(14):
        o, r, d, _ = env.step(0)
        closest = env.game.closest_goal()
        if closest is None:                                          # all goals consumed: reported as (-1, -1)
            assert d and (o[20], o[21]) == (-1, -1)
            break
        seen.append((o[20], o[21], r))
        assert (closest.x, closest.y) == (o[20], o[21])
    ys = [s[1] for s in seen]
    assert ys[0] == 100 and ys == sorted(ys) and len(set(ys)) >= 3   # switches to the next goal as they are reached
    assert sum(1 for s in seen if s[2] == 1.0) >= 2


def test_reward(env):
    env.reset(spawn_point=(300, 60), goals=[[305, 140]] + FAR[:4])
    rewards = []
    for _ in range(8):
        _, r, d, _ = env.step(0)
        rewards.append(r)
    assert rewards.count(1.0) == 1                                   # touching a goal: exactly +1, once
    assert all(r == STEP_PENALTY for r in rewards if r != 1.0)       # every other step: exactly STEP_PENALTY
    assert env.cumulative_reward == pytest.approx(sum(rewards))


def test_done_goals_reached(env):
    env.reset(spawn_point=(300, 60), goals=[[305, 100], [305, 130], [305, 160], [305, 190], [305, 220]])
    done = False
    for _ in range(30):
        _, r, done, _ = env.step(0)
        if done:
            break
    assert done and len(env.game.goals) == 0 and not env.game.colliding
    assert env.game.player.y < 600                                    # ended by the goals, not by leaving the map


def test_done_out_of_bounds(env):
    env.reset(spawn_point=(300, 500), goals=[[60, 300]] * 5)         # goals out of the way, ship near the top edge
    done, r = False, None
    for _ in range(40):
        _, r, done, _ = env.step(0)
        if done:
            break
    assert done and env.game.player.y > 600 and r == -1.0
    env.reset(spawn_point=(300, 30), goals=[[60, 300]] * 5)
    _, r, done, _ = env.step(1)
    assert not done and r == STEP_PENALTY
