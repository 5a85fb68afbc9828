"""TEST INFRASTRUCTURE — deterministic random model-context inputs (numpy legacy RandomState, which
is bit-stable across numpy versions/platforms).  Used by the golden generator and by the tests,
so that fixtures only store outputs (inputs are a recipe: dims + seed)."""
import numpy as np


def random_context(dims, seed, B=1, t_fill=None, n_agents=None, n_polys=None):
    """Model inputs in the reference's layout/dtypes (float64 / int64 like the NumPy host path).
    t_fill: number of leading window steps that exist (default all T).  n_agents: live slots (rest padded
    like select_relevant_agents: zeros, types -1).  n_polys: live polylines (rest zero, types -1)."""
    rs = np.random.RandomState(seed)
    A, T, P, NP = dims.A, dims.T, dims.P, dims.NP
    t_fill = T if t_fill is None else t_fill
    n_agents = A if n_agents is None else n_agents
    n_polys = P if n_polys is None else n_polys
    st = np.zeros((B, A, T, 8))
    st[..., 0:2] = rs.uniform(-50, 50, (B, A, T, 2))
    st[..., 2:4] = rs.uniform(-10, 10, (B, A, T, 2))
    st[..., 4] = rs.uniform(-np.pi, np.pi, (B, A, T))
    st[..., 5] = rs.uniform(4.0, 5.5, (B, A, 1))
    st[..., 6] = rs.uniform(1.8, 2.3, (B, A, 1))
    st[..., 7] = 1.0
    # a few agents vanish part-way through (existence 0 rows keep their raw values, like the reference)
    drop = rs.uniform(size=(B, A)) < 0.2
    drop_t = rs.randint(1, max(2, T), size=(B, A))
    for b in range(B):
        for a in range(A):
            if drop[b, a]:
                st[b, a, drop_t[b, a]:, 7] = 0.0
    st[:, :, t_fill:, :] = 0.0
    st[:, n_agents:] = 0.0
    types = np.zeros((B, A, 5))
    types[..., 1] = 1.0
    types[:, n_agents:] = -1.0
    goals = np.concatenate([rs.uniform(-80, 80, (B, A, 2)), rs.uniform(-10, 10, (B, A, 2)),
                            rs.uniform(-np.pi, np.pi, (B, A, 1))], -1)
    goals[:, n_agents:] = 0.0
    actions = rs.randint(0, dims.V, (B, A, T)).astype(np.float64)
    rtgs = rs.randint(0, dims.R, (B, A, T, 3)).astype(np.float64)
    t0 = rs.randint(0, dims.MAXT - T + 1)
    timesteps = np.broadcast_to((t0 + np.arange(T))[None, None, :, None], (B, A, T, 1)).astype(np.int64).copy()
    rp = np.zeros((B, P, NP, 3))
    start = rs.uniform(-60, 60, (B, P, 1, 2))
    step = rs.normal(0, 1.0, (B, P, NP, 2))
    rp[..., :2] = start + np.cumsum(step, axis=2)
    npts = rs.randint(1, NP + 1, (B, P))
    rp[..., 2] = (np.arange(NP)[None, None, :] < npts[..., None]).astype(np.float64)
    rp[..., :2] *= rp[..., 2:3]
    rt = np.zeros((B, P, 8))
    kinds = rs.randint(1, 7, (B, P))
    for b in range(B):
        rt[b, np.arange(P), kinds[b]] = 1.0
    rp[:, n_polys:] = 0.0
    rt[:, n_polys:] = -1.0
    return dict(agent_states=st, agent_types=types, goals=goals, actions=actions, rtgs=rtgs,
                timesteps=timesteps, moving_agent_mask=np.ones((B, A)),
                road_points=rp, road_types=rt)


def to_torch(d):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


def to_motion_data(d):
    """The reference's MotionData-like dict: data['agent'].x / data['map'].x (attribute access)."""
    import torch

    class _Store(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    t = to_torch(d)
    return {"agent": _Store({k: t[k] for k in ("agent_states", "agent_types", "goals", "actions", "rtgs",
                                                "timesteps", "moving_agent_mask")}),
            "map": _Store({k: t[k] for k in ("road_points", "road_types")})}


def synth_policy_buffers(scn, cfg, seed):
    """Deterministic pseudo-history for Policy buffers: agents drift along their heading; one agent per 7
    drives away fast so that it leaves the 60 m disc of its group later on."""
    rs = np.random.RandomState(seed)
    N, steps = scn.N, cfg.nocturne.steps
    w = cfg.dataset.waymo
    states = np.zeros((N, steps, 8))
    t = np.arange(steps)[None, :]
    sp = scn.speed.astype(np.float64)[:, None] * (1 + 3.0 * (np.arange(N) % 7 == 3))[:, None]
    hd = scn.heading.astype(np.float64)[:, None] + 0.01 * t
    states[..., 0] = scn.x[:, None] + np.cumsum(sp * np.cos(hd) * 0.1, axis=1)
    states[..., 1] = scn.y[:, None] + np.cumsum(sp * np.sin(hd) * 0.1, axis=1)
    states[..., 2] = sp * np.cos(hd)
    states[..., 3] = sp * np.sin(hd)
    states[..., 4] = hd
    states[..., 5] = scn.length[:, None]
    states[..., 6] = scn.width[:, None]
    states[..., 7] = 1.0
    states = states.astype(np.float32).astype(np.float64)
    actions = np.stack([rs.uniform(-12, 12, (N, steps)), rs.uniform(-0.9, 0.9, (N, steps))], -1)
    rtgs = np.stack([rs.uniform(-1, 11, (N, steps)), rs.uniform(-20, 100, (N, steps)),
                     rs.uniform(-20, 100, (N, steps))], -1)
    goals = np.repeat(scn.goals5()[:, None], steps, 1)
    timesteps = np.repeat(np.arange(steps)[None, :, None], N, 0).astype(np.float64)
    return dict(states=states, types=scn.types.copy(), actions=actions, rtgs=rtgs, goals=goals,
                timesteps=timesteps)


def dt_rtgs(rtg_bins, seed):
    """Continuous RTGs in [0, 1] (the Decision-Transformer variant) of the shape of the synthetic bins."""
    return np.random.RandomState(1000 + seed).uniform(0.0, 1.0, np.shape(rtg_bins))
