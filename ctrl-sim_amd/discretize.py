"""Action / RTG (un)discretisation helpers of the host side (float64 NumPy), same names as the reference's
RLWaymoDataset methods (datasets/rl_waymo/dataset.py:322-387) so that host code written against the reference reads
the same.  `w` is cfg.dataset.waymo."""
import numpy as np


def discretize_actions(actions, w):
    a = np.asarray(actions, np.float64)
    a0 = (np.clip(a[..., 0], w.min_accel, w.max_accel) - w.min_accel) / (w.max_accel - w.min_accel)
    a1 = (np.clip(a[..., 1], w.min_steer, w.max_steer) - w.min_steer) / (w.max_steer - w.min_steer)
    return np.round(a0 * (w.accel_discretization - 1)) * w.steer_discretization + np.round(a1 * (w.steer_discretization - 1))


def undiscretize_actions(tokens, w):
    tok = np.asarray(tokens)
    out = np.zeros(tok.shape + (2,))
    out[..., 0] = (tok // w.steer_discretization) / (w.accel_discretization - 1)
    out[..., 1] = (tok % w.steer_discretization) / (w.steer_discretization - 1)
    out[..., 0] = out[..., 0] * (w.max_accel - w.min_accel) + w.min_accel
    out[..., 1] = out[..., 1] * (w.max_steer - w.min_steer) + w.min_steer
    return out


def _rtg_ranges(w):
    return (w.min_rtg_pos, w.min_rtg_veh, w.min_rtg_road), (w.max_rtg_pos, w.max_rtg_veh, w.max_rtg_road)


def discretize_rtgs_from_raw(rtgs, w):
    """clip + scale to [0,1] (autoregressive_policy.py:73-78) then round(x * (R-1)) (dataset.py:382-387)."""
    lo, hi = _rtg_ranges(w)
    r = np.asarray(rtgs, np.float64).copy()
    for c in range(3):
        r[..., c] = np.round((np.clip(r[..., c], lo[c], hi[c]) - lo[c]) / (hi[c] - lo[c]) * (w.rtg_discretization - 1))
    return r


def undiscretize_rtgs(bins, w):
    lo, hi = _rtg_ranges(w)
    b = np.asarray(bins, np.float64)
    out = np.zeros_like(b)
    for c in range(3):
        out[..., c] = (b[..., c] / (w.rtg_discretization - 1)) * (hi[c] - lo[c]) + lo[c]
    return out


def get_tilt_logits(goal_tilt, veh_tilt, road_tilt, w):
    lin = np.linspace(0, 1, w.rtg_discretization)
    return np.stack([goal_tilt * lin, veh_tilt * lin, road_tilt * lin], axis=1)
