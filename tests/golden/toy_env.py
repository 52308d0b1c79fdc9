"""A deterministic toy single env (plain Python, Gym-0.21 protocol) used to pin the WRAPPER logic of the reference's vector
envs (DummyVecEnv.step_wait auto-reset + terminal stash, VecRecordEpisodeStatistics accumulators): the generator runs the
reference's wrappers over it, the test runs the oracle's RefVecEnv over an identical copy."""
import numpy as np


class ToyEnv:
    def __init__(self, ident, horizon):
        self.ident, self.horizon = ident, horizon
        self.observation_space = self.action_space = None
        self.episodes = -1

    def reset(self):
        self.t = 0
        self.episodes += 1
        return np.array([self.ident, self.episodes, 0.0]), {'reset_of': self.ident}

    def step(self, action):
        self.t += 1
        obs = np.array([self.ident, self.episodes, float(self.t)]) + 0.01 * float(action[0])
        rew = 0.1 * self.t * (1 + self.ident) - float(action[0]) ** 2
        done = self.t >= self.horizon or float(action[0]) > 0.9
        return obs, rew, done, {'constraint_violation': int(float(action[0]) > 0.5), 'mse': 0.5 * self.t}

    def close(self):
        pass
