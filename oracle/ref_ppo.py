"""Oracle: returns / GAE advantages, restating controllers/ppo/ppo_utils.py:428-456 (float64 NumPy).

Test infrastructure -- see ``oracle/__init__.py``.  Same loop, same update order, including the in-place
``rews += gamma * terminal_vals`` (:443) and ``vals`` extended by ``last_val`` (:441); the advantage
normalisation restates ppo.py:288 (population std, ``+ 1e-6``).
"""
import numpy as np


def compute_returns_and_advantages(rews, vals, masks, terminal_vals=0, last_val=0, gamma=0.99, use_gae=False,
                                   gae_lambda=0.95):
    rews = np.array(rews, dtype=np.float64)
    vals = np.asarray(vals, dtype=np.float64)
    masks = np.asarray(masks, dtype=np.float64)
    T, N = rews.shape[:2]
    last_val = np.broadcast_to(np.asarray(last_val, dtype=np.float64).reshape(-1, 1), (N, 1))
    rets, advs = np.zeros((T, N, 1)), np.zeros((T, N, 1))
    ret, adv = last_val, np.zeros((N, 1))
    vals = np.concatenate([vals, last_val[np.newaxis, ...]], 0)
    rews = rews + gamma * np.asarray(terminal_vals, dtype=np.float64)
    for i in reversed(range(T)):
        ret = rews[i] + gamma * masks[i] * ret
        if not use_gae:
            adv = ret - vals[i]
        else:
            td_error = rews[i] + gamma * masks[i] * vals[i + 1] - vals[i]
            adv = adv * gae_lambda * gamma * masks[i] + td_error
        rets[i] = ret
        advs[i] = adv
    return rets, advs, rews


def normalize(adv):
    return (adv - adv.mean()) / (adv.std() + 1e-6)
