"""Oracle: the reference's running-statistics normalisers, float64 NumPy.

Test infrastructure -- see ``oracle/__init__.py``.

Follows /root/reference/safe_control_gym/math_and_models/normalization.py statement by statement:
``RunningMeanStd`` (:17-62), ``MeanStdNormalizer.__call__`` (:125-133), ``RewardStdNormalizer.__call__``
(:172-189).  The reference module itself cannot be imported here (``from gym.spaces import Box``, :5; gym is
absent) and its tests hold no vectors for it, so the pins are analytic: Chan's update must reproduce
``np.mean`` / ``np.var`` of the concatenated stream (with the epsilon-count prior), see tests/test_normalization.py.

``legacy`` selects the literal reading of :188 (``self.ret[dones.astype(np.long)] = 0``: fancy indexing with the
0 / 1 values, not a mask).
"""
import numpy as np


class RunningMeanStd:
    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def update(self, arr):
        self.update_from_moments(np.mean(arr, axis=0), np.var(arr, axis=0), arr.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot_count
        m_a = self.var * self.count
        m_b = batch_var * batch_count
        m_2 = m_a + m_b + np.square(delta) * self.count * batch_count / (self.count + batch_count)
        self.mean, self.var, self.count = new_mean, m_2 / (self.count + batch_count), batch_count + self.count


class MeanStdNormalizer:
    def __init__(self, shape=(), read_only=False, clip=10.0, epsilon=1e-8):
        self.read_only = read_only
        self.rms = RunningMeanStd(shape=shape)
        self.clip, self.epsilon = clip, epsilon

    def __call__(self, x):
        x = np.asarray(x, dtype=np.float64)
        if not self.read_only:
            self.rms.update(x)
        return np.clip((x - self.rms.mean) / np.sqrt(self.rms.var + self.epsilon), -self.clip, self.clip)


class RewardStdNormalizer(MeanStdNormalizer):
    def __init__(self, gamma=0.99, read_only=False, clip=10.0, epsilon=1e-8, legacy=False):
        super().__init__((), read_only, clip, epsilon)
        self.gamma, self.ret, self.legacy = gamma, None, legacy

    def __call__(self, x, dones):
        x = np.asarray(x, dtype=np.float64)
        if not self.read_only:
            if self.ret is None:
                self.ret = np.zeros(x.shape[0])
            self.ret = self.ret * self.gamma + x
            self.rms.update(self.ret)
            if self.legacy:
                self.ret[np.asarray(dones).astype(np.int64)] = 0          # :188 as written
            else:
                self.ret[np.asarray(dones).astype(bool)] = 0              # :187 as commented
        return np.clip(x / np.sqrt(self.rms.var + self.epsilon), -self.clip, self.clip)
