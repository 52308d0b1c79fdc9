"""MI355X-native vectorised rollout engine for safe-control-gym's ``env.step()`` hot path.

``make('quadrotor', num_envs=N, device='cuda', **reference_kwargs)`` returns a batched environment
whose ``reset()`` / ``step()`` run as fused HIP kernels on gfx950.  Registration
mirrors the reference's ``safe_control_gym/envs/__init__.py:6-8``; ``cartpole`` is registered too
although the reference fork dropped that environment (parity unpinned).
"""
from safe_control_gym_amd.utils.registration import register, make, spec, get_config   # noqa: F401

register(id='quadrotor',
         entry_point='safe_control_gym_amd.envs.quadrotor:Quadrotor',
         config_entry_point='safe_control_gym_amd.envs:quadrotor_defaults.yaml')

register(id='cartpole',
         entry_point='safe_control_gym_amd.envs.cartpole:CartPole',
         config_entry_point='safe_control_gym_amd.envs:cartpole_defaults.yaml')

__version__ = '0.1.0'
