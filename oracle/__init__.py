"""CPU oracle for the safe-control-gym ``env.step()`` hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``safe_control_gym_amd/`` may import,
call, link or execute anything in this package: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and
there only as the checker, never as the thing being measured or shipped.

What it is
----------
A float64, one-environment-at-a-time restatement (plain Python / NumPy) of the
reference's step path, function for function, each citing the reference
``file:line`` it follows (paths relative to ``/root/reference``).  The batched
"vector env" here is a Python ``for`` over single-env objects, exactly like the
reference's ``DummyVecEnv``.

Modules: ``ref_model`` / ``ref_env`` / ``ref_constraints`` / ``ref_trajectory`` / ``ref_vec`` (the step path),
``philox`` (the reset / noise generator), ``ref_disturbances`` (envs/disturbances.py), ``ref_ppo`` (returns / GAE),
``ref_linearize`` (df_func Jacobians, discretisation, quadratic loss), ``ref_normalization`` (running mean / std
normalisers), ``independent`` (sympy / DOP853 cross-check), ``c/`` + ``c_port`` (plain-C port = CPU baseline).

Pinning status
--------------
* The reference cannot be imported in the build container (``casadi``,
  ``pybullet`` and ``gym`` are not installed; ordinary ``ModuleNotFoundError``,
  nothing was denied) and its test-suite holds **no** golden vectors or
  known-answer tests.
* The oracle is therefore pinned by (a) the analytic known-answer tests listed
  in SURVEY.md section 8(c) (hover fixed point, free fall, pure yaw / pitch couples,
  action-bound constants, the competition ``BoundedConstraint`` numbers ...),
  (b) an independent re-derivation of the dynamics with sympy and a
  high-accuracy ``scipy.integrate.solve_ivp`` integration
  (``oracle/independent.py``), and (c) the one importable reference module,
  ``safe_control_gym.envs.disturbances``, RUN in the build container with a stub ``np_random`` to record
  ``tests/golden/disturbances.json`` (``tests/golden/make_golden_disturbances.py``); ``ref_disturbances``
  reproduces those recordings to 1e-13.
* (d) **the reference's own NumPy code, executed**: ``tests/golden/make_golden_reference_numpy.py`` parses the cited reference files
  with ``ast``, compiles 64 definitions / statement runs unchanged and runs them with NumPy and plain attribute holders (no module is
  stubbed; nothing that needs casadi / pybullet / gym is executed); ``tests/test_reference_pins.py`` replays this package against the
  recorded inputs and outputs (<= 1e-12).  Covered: motor map, trajectory generators and projection, X_GOAL assembly, every
  Constraint class, bounds, randomisation, the step glue, the vector-env wrappers, the PPO buffer and GAE, the normalisers,
  discretisation, the vehicle constants (URDF parser), the rigid-body terms of the 3-D model (``BaseAviary._dynamics`` /
  ``_physics``), ground effect and drag, the Runge-Kutta statements of ``rk_discrete``, the quadratic cost statement, the default
  tables.  NOT covered (CasADi builds them): the Euler-rate kinematics as the reference writes them (checked as a property of the
  pinned rotation instead), the Jacobians, ``fd_func``.
* ``ref_linearize`` differentiates the reference's matrix-form dynamics symbolically (the sympy analogue of the
  reference's ``cs.jacobian``) and is cross-checked by central differences; ``ref_normalization`` restates
  ``math_and_models/normalization.py`` (not importable: it imports gym) and is pinned by the identity "Chan's
  update == statistics of the concatenated stream".
* PyBullet and CVODES numerics: **parity unpinned** (third-party native code,
  absent here, no reference fixture covers them).  CartPole: **parity
  unpinned** (no CartPole environment exists in this fork of the reference).
"""
