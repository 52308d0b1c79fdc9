"""Batched discretisation of linear systems -- device mirror of ``discretize_linear_system``
(/root/reference/safe_control_gym/controllers/mpc/mpc_utils.py:24-54; the same function lives in
controllers/lqr/lqr_utils.py:39-60).

``exact=False``: ``Ad = I + A dt``, ``Bd = B dt`` (a fused option of ``scg_linearize`` as well).  ``exact=True``:
``[[Ad, Bd], [0, I]] = expm([[A, B], [0, 0]] dt)`` for the whole batch in one launch of ``scg_discretize_exact``
(csrc/scg_expm.hip: Taylor + scaling and squaring on ``v_mfma_f32_16x16x4_f32``).  Inputs: float32 device tensors
``A [N, nx, nx]``, ``B [N, nx, nu]`` (what ``env.linearize`` returns), or single NumPy matrices (returned as NumPy).
"""
import numpy as np
import torch

from safe_control_gym_amd import _capi


def _batch(M, dev):
    single = not isinstance(M, torch.Tensor) or M.ndim == 2
    t = torch.as_tensor(np.asarray(M, dtype=np.float32) if not isinstance(M, torch.Tensor) else M, dtype=torch.float32, device=dev)
    if t.ndim == 2:
        t = t[None]
    if t.ndim != 3:
        raise ValueError('[ERROR] expected a matrix or a batch of matrices, got shape %s' % (tuple(t.shape),))
    return t.contiguous(), single


def discretize_exact(A, B, dt, z0=None, w=None, device='cuda', want_matrices=True):
    """-> (Ad [N, nx, nx], Bd [N, nx, m], z1 [N, nx] or None).  ``z0`` ([N, nx]) asks for ``z1 = Ad z0 + Bd w``
    (``w`` [N, m], default ones) from the same launch."""
    dev = A.device if isinstance(A, torch.Tensor) else torch.device(device)
    if dev.type != 'cuda':
        raise _capi.ScgError('discretize_exact runs on a ROCm device only (got %r); there is no CPU path' % (dev,))
    A, _ = _batch(A, dev)
    B, _ = _batch(B, dev)
    n, nx, m = A.shape[0], A.shape[1], B.shape[2]
    if A.shape[2] != nx or B.shape[:2] != (n, nx):
        raise ValueError('[ERROR] A must be (N, nx, nx) and B (N, nx, m); got %s and %s' % (tuple(A.shape), tuple(B.shape)))
    if nx + m > 16:
        raise ValueError('[ERROR] exact discretisation handles nx + m <= 16 (got %d + %d)' % (nx, m))
    lib = _capi.load()
    Ad = torch.empty_like(A) if want_matrices else None
    Bd = torch.empty_like(B) if want_matrices else None
    z1 = None
    if z0 is not None:
        z0 = torch.as_tensor(z0, dtype=torch.float32, device=dev).reshape(n, nx).contiguous()
        if w is not None:
            w = torch.as_tensor(w, dtype=torch.float32, device=dev).reshape(n, m).contiguous()
        z1 = torch.empty((n, nx), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _capi.check(lib.scg_discretize_exact(_capi.ptr(A), _capi.ptr(B), nx, m, float(dt), _capi.ptr(Ad), _capi.ptr(Bd),
                                             _capi.ptr(z0), _capi.ptr(w), _capi.ptr(z1), n, _capi.current_stream(dev)),
                    'scg_discretize_exact')
    return Ad, Bd, z1


def discretize_linear_system(A, B, dt, exact=False):
    """Same signature and meaning as the reference's (mpc_utils.py:24-54), for one system or a batch."""
    if not exact:
        if isinstance(A, torch.Tensor):
            eye = torch.eye(A.shape[-1], dtype=A.dtype, device=A.device)
            return eye + A * dt, B * dt
        A, B = np.asarray(A, dtype=np.float64), np.asarray(B, dtype=np.float64)
        return np.eye(A.shape[-1]) + A * dt, B * dt
    numpy_in = not isinstance(A, torch.Tensor)
    single = (np.ndim(A) if numpy_in else A.ndim) == 2
    Ad, Bd, _ = discretize_exact(A, B, dt)
    if single:
        Ad, Bd = Ad[0], Bd[0]
    if numpy_in:
        return Ad.double().cpu().numpy(), Bd.double().cpu().numpy()
    return Ad, Bd
