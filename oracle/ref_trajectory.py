"""Oracle: reference trajectories and the X_GOAL / U_GOAL tables (float64, per-sample loop).

Test infrastructure -- see ``oracle/__init__.py``.

Reference anchors (relative to /root/reference/safe_control_gym):
  * envs/benchmark_env.py:465-519   _generate_trajectory (np.arange sampling, per-sample loop)
  * envs/benchmark_env.py:521-566   _get_coordinates (plane placement)
  * envs/benchmark_env.py:568-674   _figure8 / _circle / _square closed forms
  * math_and_models/transformations.py:52-105   projection_matrix (orthogonal branch)
  * math_and_models/transformations.py:107-123  transform_trajectory (the homogeneous '1' is
    appended to velocities too, so the plane offset is added to them as well -- restated
    as is)
  * envs/gym_pybullet_drones/quadrotor.py:261-329  X_GOAL / U_GOAL / X_EQ / U_EQ assembly
"""
import math

import numpy as np


def _figure8(t, period, scaling):
    w = 2.0 * np.pi / period                                    # benchmark_env.py:587
    a = scaling * np.sin(w * t)
    b = scaling * np.sin(w * t) * np.cos(w * t)
    a_dot = scaling * w * np.cos(w * t)
    b_dot = scaling * w * (np.cos(w * t) ** 2 - np.sin(w * t) ** 2)
    return a, b, a_dot, b_dot


def _circle(t, period, scaling):
    w = 2.0 * np.pi / period                                    # benchmark_env.py:613
    a = scaling * np.cos(w * t)
    b = scaling * np.sin(w * t)
    a_dot = -scaling * w * np.sin(w * t)
    b_dot = scaling * w * np.cos(w * t)
    return a, b, a_dot, b_dot


def _square(t, period, scaling):
    seg_period = period / 4.0                                   # benchmark_env.py:640-649
    speed = scaling / seg_period
    cycle_time = t % period
    seg_time = cycle_time % seg_period
    seg_index = int(np.floor(cycle_time / seg_period))
    seg_pos = speed * seg_time
    if seg_index == 0:
        return 0.0, seg_pos, 0.0, speed
    if seg_index == 1:
        return -seg_pos, scaling, -speed, 0.0
    if seg_index == 2:
        return -scaling, scaling - seg_pos, 0.0, -speed
    if seg_index == 3:
        return -scaling + seg_pos, 0.0, speed, 0.0
    raise ValueError('unreachable segment index')


_TRAJ = {'figure8': _figure8, 'circle': _circle, 'square': _square}


def generate_trajectory(traj_type='figure8', traj_length=10.0, num_cycles=1, traj_plane='xy',
                        position_offset=(0, 0), scaling=1.0, sample_time=0.01):
    """benchmark_env.py:465-519.  Returns (pos (T,3), vel (T,3), speed (T,1))."""
    if traj_type not in _TRAJ:
        raise ValueError('Trajectory type should be one of [circle, square, figure8].')
    period = traj_length / num_cycles
    axes = ['x', 'y', 'z']
    if traj_plane[0] in axes and traj_plane[1] in axes and traj_plane[0] != traj_plane[1]:
        ia, ib = axes.index(traj_plane[0]), axes.index(traj_plane[1])
    else:
        raise ValueError('Trajectory plane should be in form of ab, where a and b can be {x, y, z}.')
    times = np.arange(0, traj_length, sample_time)              # :504
    pos = np.zeros((len(times), 3))
    vel = np.zeros((len(times), 3))
    speed = np.zeros((len(times), 1))
    for k, t in enumerate(times):
        a, b, a_dot, b_dot = _TRAJ[traj_type](t, period, scaling)
        pos[k, ia] = a + position_offset[0]                     # :562-565
        vel[k, ia] = a_dot
        pos[k, ib] = b + position_offset[1]
        vel[k, ib] = b_dot
        speed[k] = np.linalg.norm(vel[k])
    return pos, vel, speed


def projection_matrix(point, normal):
    """transformations.py:80-105, orthogonal-projection branch only (the one the env uses)."""
    M = np.identity(4)
    point = np.array(point[:3], dtype=np.float64)
    n = np.array(normal[:3], dtype=np.float64)
    n = n / math.sqrt(np.dot(n, n))                             # unit_vector, :36-38
    M[:3, :3] -= np.outer(n, n)
    M[:3, 3] = np.dot(point, n) * n
    return M


def transform_trajectory(pos, vel, point, normal):
    """transformations.py:107-123."""
    M = projection_matrix(point, normal)
    aug_pos = np.concatenate([pos, np.ones((pos.shape[0], 1))], -1)
    aug_vel = np.concatenate([vel, np.ones((vel.shape[0], 1))], -1)
    return (aug_pos @ M.T)[:, :3], (aug_vel @ M.T)[:, :3]


DEFAULT_TASK_INFO = {                                           # quadrotor.py:136-146
    'stabilization_goal': [0, 1],
    'stabilization_goal_tolerance': 0.05,
    'trajectory_type': 'circle',
    'num_cycles': 1,
    'trajectory_plane': 'zx',
    'trajectory_position_offset': [0.5, 0],
    'trajectory_scale': -0.5,
    'proj_point': [0, 0, 0.5],
    'proj_normal': [0, 1, 1],
}


def build_x_goal(quad_type, task, task_info, episode_len_sec, ctrl_timestep):
    """quadrotor.py:263-325.  Stabilisation: (nx,) vector; tracking: (T, nx) table."""
    ti = task_info
    if task == 'stabilization':
        g = ti['stabilization_goal']
        if quad_type == 1:
            return np.hstack([g[1], 0.0])
        if quad_type == 2:
            return np.hstack([g[0], 0.0, g[1], 0.0, 0.0, 0.0])
        return np.hstack([g[0], 0.0, g[1], 0.0, g[2], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    pos, vel, _ = generate_trajectory(traj_type=ti['trajectory_type'],
                                      traj_length=episode_len_sec,
                                      num_cycles=ti['num_cycles'],
                                      traj_plane=ti['trajectory_plane'],
                                      position_offset=ti['trajectory_position_offset'],
                                      scaling=ti['trajectory_scale'],
                                      sample_time=ctrl_timestep)
    T = pos.shape[0]
    z = np.zeros(T)
    if quad_type == 1:
        return np.vstack([pos[:, 2], vel[:, 2]]).transpose()
    if quad_type == 2:
        return np.vstack([pos[:, 0], vel[:, 0], pos[:, 2], vel[:, 2], z, z]).transpose()
    pos_t, vel_t = transform_trajectory(pos, vel, ti['proj_point'], ti['proj_normal'])
    return np.vstack([pos_t[:, 0], vel_t[:, 0], pos_t[:, 1], vel_t[:, 1], pos_t[:, 2], vel_t[:, 2],
                      z, z, z, z, z, z]).transpose()
