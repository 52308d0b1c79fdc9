"""The ``VecEnv`` protocol (reference: env_wrappers/vectorized_env/vec_env.py:16-227) over a batched env."""


def _must_override(name):
    def method(self, *args, **kwargs):
        raise NotImplementedError('%s.%s()' % (type(self).__name__, name))
    method.__name__ = name
    return method


class VecEnv:
    """Batched-environment protocol: one ``step(actions[N, nu])`` advances N envs (vec_env.py:16-162).

    Sub-classes provide ``reset``, ``step_async``, ``step_wait``, ``get_attr``, ``set_attr``, ``env_method``.
    """
    closed, viewer = False, None
    metadata = {'render.modes': []}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space

    reset = _must_override('reset')
    step_async = _must_override('step_async')
    step_wait = _must_override('step_wait')
    get_attr = _must_override('get_attr')
    set_attr = _must_override('set_attr')
    env_method = _must_override('env_method')

    def step(self, actions):
        """Synchronous step = step_async + step_wait (vec_env.py:83-90)."""
        self.step_async(actions)
        return self.step_wait()

    def render(self, mode='human'):
        """vec_env.py:96-112 tiles the envs' camera images; the batched engine has no renderer (as ``env.render``)."""
        raise NotImplementedError('[ERROR] the batched engine does not render.')

    def get_images(self):
        raise NotImplementedError('[ERROR] the batched engine does not render.')

    def get_viewer(self):
        raise NotImplementedError('[ERROR] the batched engine does not render.')

    def close_extras(self):
        """Hook for sub-classes; runs once from close()."""

    def close(self):
        if not self.closed:
            self.close_extras()
            self.closed = True

    @property
    def unwrapped(self):
        return self.venv.unwrapped if isinstance(self, VecEnvWrapper) else self

    def _get_indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        return [indices] if isinstance(indices, int) else indices


class DeviceVecEnv(VecEnv):
    """All ``num_envs`` environments are one batched env; ``step_async`` enqueues the kernel on the
    current HIP stream and ``step_wait`` hands back device tensors without synchronising."""

    def __init__(self, env):
        self.env = env
        super().__init__(env.num_envs, env.observation_space, env.action_space)
        self._pending = None

    def reset(self):
        out = self.env.reset()
        if isinstance(out, tuple):
            return out
        return out, {'n': None}

    def step_async(self, actions):
        self._pending = self.env.step(actions)

    def step_wait(self):
        out, self._pending = self._pending, None
        return out

    def capture_loop(self, body, steps, warmup=2):
        """``steps`` iterations of ``body(k)`` (policy kernels + this venv's step calls) recorded in one HIP graph: see
        ``safe_control_gym_amd.envs.graph_loop.capture_loop``."""
        from safe_control_gym_amd.envs.graph_loop import capture_loop
        return capture_loop([self.env], body, steps, warmup)

    def close_extras(self):
        self.env.close()

    def get_attr(self, attr_name, indices=None):
        """Attributes are shared by the whole batch: one value per requested index."""
        return [getattr(self.env, attr_name) for _ in self._get_indices(indices)]

    def set_attr(self, attr_name, values, indices=None):
        idx = list(self._get_indices(indices))
        assert len(idx) == len(values)
        if any(v is not values[0] and v != values[0] for v in values):
            raise ValueError('[ERROR] the batched engine shares attributes across the batch; values differ.')
        setattr(self.env, attr_name, values[0])

    def env_method(self, method_name, method_args=None, method_kwargs=None, indices=None):
        idx = list(self._get_indices(indices))
        args = method_args[0] if method_args else []
        kwargs = method_kwargs[0] if method_kwargs else {}
        res = getattr(self.env, method_name)(*args, **kwargs)
        return [res for _ in idx]

    def get_env_random_state(self):
        return [{'seed': self.env._seed, 'serial': self.env._serial}]

    def set_env_random_state(self, worker_random_states):
        st = worker_random_states[0]
        self.env.seed(st['seed'])
        self.env._serial = st['serial']


class PipelinedDeviceVecEnv(VecEnv):
    """The batch cut into ``P`` contiguous shards that advance on ``P`` launch queues -- the device analogue of
    ``SubprocVecEnv`` splitting its env list over worker processes (subproc_vec_env.py:28: ``np.array_split``; ``step_async``
    sends every worker its chunk and returns, ``step_wait`` collects, :50-63).

    Every shard is a batched env of its own (``env_index_base`` = its first global env index, so the Philox streams -- and
    therefore every number -- are those of the unsplit batch, bit for bit) with its own HIP stream.  The point is the
    PER-SHARD protocol, for a caller with a policy in the loop::

        obs = [venv.reset(pipeline=p) for p in range(venv.num_pipelines)]
        while ...:
            for p in range(venv.num_pipelines):
                venv.step_async(policy(obs[p]), pipeline=p)      # returns at once; the kernel runs on shard p's queue
                ...                                             # the other shard's policy / bookkeeping overlaps it
            for p in range(venv.num_pipelines):
                obs[p], rew, done, info = venv.step_wait(pipeline=p)

    (or, software-pipelined, ``step_wait(p)`` -> ``policy`` -> ``step_async(p)`` per shard in turn): one shard's kernel ramp-up
    / drain and its policy overlap the other shard's kernel body, which is where the two-queue rollout's rate comes from
    (a lone dependent launch per step leaves ~5 us of every ~28 idle at 2^20 envs).  Without ``pipeline=`` the calls act on
    the whole batch -- ``actions [N, nu]`` is split by rows (zero copy), the outputs are concatenated (one copy: prefer the
    per-shard form in a hot loop)."""

    def __init__(self, envs):
        self.envs = list(envs)
        first = self.envs[0]
        super().__init__(sum(e.num_envs for e in self.envs), first.observation_space, first.action_space)
        self.num_pipelines = len(self.envs)
        self.bounds, lo = [], 0
        for e in self.envs:
            self.bounds.append((lo, lo + e.num_envs))
            lo += e.num_envs

    @property
    def env(self):
        """The first shard: what attribute reads that are the same for every shard (spaces, X_GOAL, symbolic ...) go to."""
        return self.envs[0]

    @staticmethod
    def _cat(parts):
        import torch
        if isinstance(parts[0], dict):
            return {k: PipelinedDeviceVecEnv._cat([q[k] for q in parts]) for k in parts[0]}
        return torch.cat(parts, dim=0)

    def reset(self, pipeline=None):
        if pipeline is not None:
            out = self.envs[pipeline].reset()
            return out if isinstance(out, tuple) else (out, {'n': None})
        outs = [e.reset() for e in self.envs]
        if isinstance(outs[0], tuple):
            return self._cat([o[0] for o in outs]), outs[0][1]
        return self._cat(outs), {'n': None}

    def step_async(self, actions, pipeline=None):
        if pipeline is not None:
            return self.envs[pipeline].step_async(actions)
        for e, (lo, hi) in zip(self.envs, self.bounds):
            e.step_async(actions[lo:hi])

    def step_wait(self, pipeline=None):
        if pipeline is not None:
            return self.envs[pipeline].step_wait()
        from safe_control_gym_amd.envs.benchmark_env import LazyInfoList
        outs = [e.step_wait() for e in self.envs]
        obs, rew, done = (self._cat([o[q] for o in outs]) for q in range(3))
        info = {k: self._cat([o[3][k] for o in outs]) for k in outs[0][3] if k != 'n'}
        if 'n' in outs[0][3]:
            info['n'] = LazyInfoList(dict(info), done)        # (a copy: info -> info['n'] -> info would be a reference cycle)
        return obs, rew, done, info

    def step(self, actions, pipeline=None):
        self.step_async(actions, pipeline)
        return self.step_wait(pipeline)

    def pipeline_stream(self, pipeline):
        """Shard ``pipeline``'s own launch queue as a torch stream (run its policy there and no event is needed)."""
        return self.envs[pipeline].pipeline_stream

    def capture_loop(self, body, steps, warmup=2):
        """``steps`` iterations of ``body(k)`` (policy kernels + this venv's step calls) recorded in one HIP graph: see
        ``safe_control_gym_amd.envs.graph_loop.capture_loop``."""
        from safe_control_gym_amd.envs.graph_loop import capture_loop
        return capture_loop(self.envs, body, steps, warmup)

    def close_extras(self):
        for e in self.envs:
            e.close()

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.env, attr_name) for _ in self._get_indices(indices)]

    def set_attr(self, attr_name, values, indices=None):
        idx = list(self._get_indices(indices))
        assert len(idx) == len(values)
        if any(v is not values[0] and v != values[0] for v in values):
            raise ValueError('[ERROR] the batched engine shares attributes across the batch; values differ.')
        for e in self.envs:
            setattr(e, attr_name, values[0])

    def env_method(self, method_name, method_args=None, method_kwargs=None, indices=None):
        idx = list(self._get_indices(indices))
        args = method_args[0] if method_args else []
        kwargs = method_kwargs[0] if method_kwargs else {}
        res = [getattr(e, method_name)(*args, **kwargs) for e in self.envs]
        return [res[0] for _ in idx]

    def get_env_random_state(self):
        return [{'seed': e._seed, 'serial': e._serial} for e in self.envs]

    def set_env_random_state(self, worker_random_states):
        for e, st in zip(self.envs, worker_random_states):
            e.seed(st['seed'])
            e._serial = st['serial']


class VecEnvWrapper(VecEnv):
    """vec_env.py:165-227."""

    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        super().__init__(num_envs=venv.num_envs, observation_space=observation_space or venv.observation_space,
                         action_space=action_space or venv.action_space)

    def step_async(self, actions):
        self.venv.step_async(actions)

    def reset(self):
        return self.venv.reset()

    def step_wait(self):
        return self.venv.step_wait()

    def close(self):
        return self.venv.close()

    def get_attr(self, attr_name, indices=None):
        return self.venv.get_attr(attr_name, indices)

    def set_attr(self, attr_name, values, indices=None):
        return self.venv.set_attr(attr_name, values, indices)

    def env_method(self, method_name, method_args=None, method_kwargs=None, indices=None):
        return self.venv.env_method(method_name, method_args, method_kwargs, indices)

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError("attempted to get missing private attribute '{}'".format(name))
        return getattr(self.venv, name)
