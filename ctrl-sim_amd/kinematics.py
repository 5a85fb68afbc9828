"""Host-side kinematics helpers of the rollout driver.

`bicycle_backward` is the inverse bicycle model the reference uses for log-replay (uncontrolled /
history-step vehicles): nocturne/bicycle_model.py:51-109 called from evaluators/evaluator.py:160-193.
Vectorised float64 NumPy; pinned by tests/golden/bicycle_backward.npz.
"""
import numpy as np


def angle_sub(current, target):
    """utils/geometry.py:3-12 (array form)."""
    d = (np.asarray(target) - np.asarray(current)) % (2 * np.pi)
    return np.where(d > np.pi, -(2 * np.pi - d), d)


def bicycle_backward(nxt, prev, dt):
    """nxt [n,5] = (x, y, theta, vel, L) of the next state, prev [n,4] = (x, y, theta, vel) of the current one.
    Returns (accel[n], steer[n]) with steer clipped to +-0.7 and NaN -> 0."""
    nxt = np.asarray(nxt, np.float64)
    prev = np.asarray(prev, np.float64)
    accel = (nxt[:, 3] - prev[:, 3]) / dt
    w = angle_sub(prev[:, 2], nxt[:, 2]) / dt
    C = 2.0 * nxt[:, 4] * w / (nxt[:, 3] + prev[:, 3] + 1e-10)
    with np.errstate(invalid="ignore", divide="ignore"):
        steer = np.arctan(2.0 * C / np.sqrt(4 - C ** 2))
    steer = np.where(np.isnan(steer), 0.0, steer)
    return accel, np.clip(steer, -0.7, 0.7)
