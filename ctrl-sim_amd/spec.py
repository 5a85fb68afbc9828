"""Frozen constants of the CtRL-Sim closed-loop rollout path.

Every value is taken from the reference's Hydra YAML groups (citations are relative to
/root/reference): cfgs/config.yaml:44-46 (nocturne), cfgs/dataset/waymo/base.yaml,
cfgs/model/{base,ctrl_sim}.yaml, cfgs/policy/{base,ctrl_sim}.yaml.  Hydra/omegaconf are not
part of this build; `make_cfg()` returns an attribute-style namespace with `.copy()` and
`__getitem__`, which is all the reference's Policy/Encoder/Decoder/RLWaymoDataset read.
"""
from __future__ import annotations

import copy


class Cfg(dict):
    """Attribute + item access, `.copy()` is deep (Policy.__init__ calls cfg.copy(), policy.py:23)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return copy.deepcopy(self)

    def get(self, k, default=None):
        return dict.get(self, k, default)


def _wrap(d):
    return Cfg({k: _wrap(v) if isinstance(v, dict) else v for k, v in d.items()})


# cfgs/dataset/waymo/base.yaml
WAYMO = dict(
    dataset_path="", preprocess=True, preprocess_dir="/tmp/ctrlsim_preprocess",
    train_context_length=32, num_agent_types=5, num_road_types=8, map_attr=2, k_attr=7,
    agent_dist_threshold=60.0, map_dist_threshold=100.0, max_timestep=90,
    parked_car_velocity_threshold=0.05,
    max_accel=10.0, min_accel=-10.0, max_steer=0.7, min_steer=-0.7,
    max_veh_veh_distance=15.0, dist_to_road_edge_scaling_factor=15.0,
    veh_veh_collision_rew_multiplier=10.0, veh_edge_collision_rew_multiplier=10.0,
    pos_goal_shaped_min=0, pos_goal_shaped_max=0.2, pos_target_achieved_rew_multiplier=10.0,
    moving_threshold=0.05,
    min_rtg_pos=0, max_rtg_pos=10, min_rtg_yaw=0, max_rtg_yaw=110, min_rtg_vel=0, max_rtg_vel=110,
    min_rtg_veh=-10, max_rtg_veh=90, min_rtg_road=-10, max_rtg_road=90,
    max_num_agents=24, max_num_road_polylines=200, max_num_road_pts_per_polyline=100,
    accel_discretization=20, steer_discretization=50, rtg_discretization=350,
    preprocess_real_data=False, preprocess_simulated_data=False,
    simulated_dataset="", simulated_dataset_preprocessed_dir="",
    goal_dim=5, remove_shaped_goal=True, remove_shaped_veh_reward=False, remove_shaped_edge_reward=False,
)

# cfgs/model/base.yaml + cfgs/model/ctrl_sim.yaml
MODEL = dict(
    hidden_dim=256, map_attr=3, num_road_types=8, no_actions=False, num_heads=8,
    num_reward_components=3, dim_feedforward=1024, dropout=0.1, state_dim=12, use_map=True,
    goal_dropout=0.1, max_pool_map=True, supervise_moving=True, predict_rtg=True,
    attend_own_return_action=False, trajeglish=False, il=False, ctg_plus_plus=False,
    decision_transformer=False,
    num_transformer_encoder_layers=2, num_decoder_layers=4, predict_future_states=True,
    local_frame_predictions=False, loss_action_coef=1.0, encode_initial_state=True,
)

# cfgs/config.yaml:44-46 and the rew_cfg block used by utils/sim.py:83-141
NOCTURNE = dict(
    steps=90, dt=0.1, history_steps=10, collision_fix=True,
    rew_cfg=dict(
        shared_reward=False, goal_tolerance=0.5, reward_scaling=1.0, collision_penalty=0,
        shaped_goal_distance_scaling=0.2, shaped_goal_distance=True, goal_distance_penalty=False,
        goal_achieved_bonus=0, position_target=True, position_target_tolerance=1.0,
        speed_target=True, speed_target_tolerance=1.0, heading_target=True,
        heading_target_tolerance=0.3,
    ),
)

# cfgs/policy/base.yaml + cfgs/policy/ctrl_sim.yaml
POLICY = dict(
    run_name="ctrl_sim", model_path="", veh_veh_tilt=0, veh_edge_tilt=0, goal_tilt=0,
    action_temperature=1.0, nucleus_sampling=False, nucleus_threshold=0.8,
    use_rtg=True, predict_rtgs=True, discretize_rtgs=True, real_time_rewards=False,
    privileged_return=False, max_return=False, min_return=False, model="ctrl_sim",
)

# cfgs/eval/base.yaml (only the fields the rollout driver reads)
EVAL = dict(
    seed=0, eval_mode="multi_agent", multi_agent_eval_threshold=8, history_steps=10, num_files_to_evaluate=1000,
    partitions=1, partition=0, visualize=False, verbose=False,
    # one_agent / two_agent vehicle selection (find_interesting_agent / find_interesting_pair); the reference's base.yaml ships
    # eval_mode: one_agent, the rollout metric of BASELINE.json is the multi_agent one
    interesting_traj_len_threshold=60, interesting_goal_dist_threshold=10, interesting_timestep_diff_threshold=20,
)

# cfgs/eval_planner_adversary/base.yaml with cfgs/policy/ctrl_sim_{planner,adversary}.yaml mounted at .planner / .adversary
EVAL_PLANNER_ADVERSARY = dict(
    history_steps=10, verbose=True, seed=0, visualize=False, num_files_to_evaluate=1000,
    planner=dict(POLICY, goal_tilt=10, veh_veh_tilt=10, veh_edge_tilt=10),
    adversary=dict(POLICY, goal_tilt=0, veh_veh_tilt=-10, veh_edge_tilt=0),
)


def make_cfg(**overrides):
    """Build the attribute-style cfg. `overrides` are dotted keys with '__' separators,
    e.g. make_cfg(dataset__waymo__max_num_agents=4, nocturne__steps=20)."""
    cfg = _wrap(dict(
        dataset=dict(waymo=dict(WAYMO)), model=dict(MODEL), nocturne=copy.deepcopy(NOCTURNE),
        eval=dict(EVAL, policy=dict(POLICY)), eval_planner_adversary=copy.deepcopy(EVAL_PLANNER_ADVERSARY),
        dataset_root="", nocturne_waymo_val_folder="",
    ))
    for k, v in overrides.items():
        node = cfg
        parts = k.split("__")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


# Model options the HIP path is built for ("Frozen constants": cfgs/model/base.yaml:10-19, ctrl_sim.yaml:4-9).  A
# checkpoint whose cfg says otherwise describes a DIFFERENT network (another mask, other token rows, no map tokens): the kernels would run
# and return logits of the wrong model, so the model layer refuses it by name instead.
_FROZEN_MODEL_FLAGS = (
    # (key, required value, what the other value changes in the reference)
    # (use_map = False, encode_initial_state = False and no_actions = True are BUILT since round 6 — ctrlsim_dims.flags, Dims.FLAGS below — for
    #  the CtRL-Sim token layout; check_supported refuses them with a baseline layout and the two scene switches together)
    # (local_frame_predictions is NOT here: it selects the TRAINING target of the predict_future_states head — world or agent frame,
    #  models/ctrl_sim.py:114,151 — and nothing in the forward pass reads it: a checkpoint trained either way is the same network)
    ("ctg_plus_plus", False, "the diffusion baseline is another model (models/ctg_plus_plus.py)"),
    ("hidden_dim", 256, "the kernels are built for 256-wide rows (csrc/common.h: DM)"),
    ("num_heads", 8, "the kernels are built for 8 heads of 32 (csrc/common.h: NHEAD, HD)"),
    ("num_reward_components", 3, "three return components per token (csrc/sample.hip)"),
    ("map_attr", 3, "road points are (x, y, exist) rows (modules/map_encoder.py:18,41; csrc/map_encoder.hip)"),
    ("num_road_types", 8, "one-hot road types of width 8 (modules/map_encoder.py:23; csrc/forward.hip: in_mlp<8>)"),
    ("state_dim", 12, "state rows of 7 kinematic values + 5 type flags (modules/encoder.py:21; csrc/context.hip)"),
)


def model_flags(cfg):
    """ctrlsim_dims.flags (include/ctrlsim.h) of a cfg: 1 = no_actions, 2 = use_map False, 4 = encode_initial_state False
    (cfgs/model/base.yaml:4,10; ctrl_sim.yaml:9; modules/encoder.py:18,84,129-130,155-170)."""
    m = cfg.model
    return (1 if bool(m.get("no_actions", False)) else 0) | (0 if bool(m.get("use_map", True)) else 2) | \
        (0 if bool(m.get("encode_initial_state", True)) else 4)


def check_supported(cfg):
    """Raise NotImplementedError naming every model option of `cfg` the HIP path does not implement (see _FROZEN_MODEL_FLAGS).
    `predict_rtg` must be on for the CtRL-Sim model and is ignored (as in the reference configs) for the IL / Trajeglish / DT baselines;
    at most one of il / trajeglish / decision_transformer may be set."""
    m = cfg.model
    bad = [f"model.{k} = {m.get(k)!r} (built for {want!r}: {why})" for k, want, why in _FROZEN_MODEL_FLAGS
           if k in m and m.get(k) != want]
    variants = [k for k in ("il", "trajeglish", "decision_transformer") if bool(m.get(k, False))]
    if len(variants) > 1:
        bad.append(f"model.{' and model.'.join(variants)} are set together (one baseline at a time: cfgs/model/{{il,trajeglish,dt}}.yaml)")
    if variants and bool(m.get("attend_own_return_action", False)):
        bad.append("model.attend_own_return_action with a baseline token layout (utils/train_utils.py:114-129 assumes the three CtRL-Sim token "
                   "types: `type_idx_j = index_j % 3`)")
    flags = model_flags(cfg)
    if flags and variants:
        bad.append(f"model.{variants[0]} with use_map = False / encode_initial_state = False / no_actions = True (built for the CtRL-Sim token layout only)")
    if (flags & 6) == 6:
        bad.append("model.use_map = False together with model.encode_initial_state = False (the reference has no scene memory to build then: "
                   "modules/encoder.py:155-170 reads an undefined tensor)")
    if not variants and not bool(m.get("predict_rtg", True)):
        bad.append("model.predict_rtg = False with the CtRL-Sim token layout (the rollout's first pass reads the return head: "
                   "policies/autoregressive_policy.py:201-221)")
    if bad:
        raise NotImplementedError("the HIP rollout path does not implement this model configuration:\n  " + "\n  ".join(bad))


class Dims:
    """Shape constants of one model context, derived from a cfg (SURVEY.md conventions)."""

    def __init__(self, cfg):
        w, m = cfg.dataset.waymo, cfg.model
        self.A = int(w.max_num_agents)            # agents per context
        self.T = int(w.train_context_length)      # context steps
        self.K = 3                                # token types: state, rtg, action
        # model variant (cfgs/model/{ctrl_sim,il,trajeglish,dt}.yaml): IL drops the rtg tokens, Trajeglish keeps only the
        # action tokens; on the device both keep the 3-slot token layout with the dropped types dead as attention keys
        # 3 = Decision Transformer (cfgs/model/dt.yaml): continuous RTG embeddings, token order (rtg, state, action)
        self.VARIANT = 1 if bool(m.get("il", False)) else (2 if bool(m.get("trajeglish", False)) else
                                                            (3 if bool(m.get("decision_transformer", False)) else 0))
        # cfg.model.attend_own_return_action (cfgs/model/base.yaml:15, default False; utils/train_utils.py:114-129): other agents' return /
        # action tokens of EARLIER timesteps are hidden.  Built in round 6 as mask mode 5 of the in-kernel-mask attention path (plain
        # 24-slot contexts, full recompute every step: engine.py)
        self.MASK_OWN = bool(m.get("attend_own_return_action", False)) and self.VARIANT == 0
        self.FLAGS = model_flags(cfg) if self.VARIANT == 0 else 0
        self.L = self.A * self.T * self.K         # decoder tokens
        self.P = int(w.max_num_road_polylines)
        self.NP = int(w.max_num_road_pts_per_polyline)
        self.D = int(m.hidden_dim)
        self.H = int(m.num_heads)
        self.F = int(m.dim_feedforward)
        self.NA = int(w.accel_discretization)
        self.NS = int(w.steer_discretization)
        self.V = self.NA * self.NS                # action vocabulary
        self.R = int(w.rtg_discretization)        # rtg bins per component
        self.C = int(m.num_reward_components)
        self.NE = int(m.num_transformer_encoder_layers)
        self.ND = int(m.num_decoder_layers)
        self.MAXT = int(w.max_timestep)
        self.GOAL = int(w.goal_dim)
        self.STATE = int(m.state_dim)
        self.FUT = self.T * 2
        self.M = self.P + self.A                  # scene-encoder / memory tokens

    def __repr__(self):
        return "Dims(" + ", ".join(f"{k}={v}" for k, v in self.__dict__.items()) + ")"


# Placeholder tokens that the reference's buffers produce for "not yet written" rows:
# zero action -> discretize_actions -> round(9.5)*50 + round(24.5) = 10*50+24 (half-to-even), and
# zero rtg -> clip/normalise -> bins (0, round(34.9), round(34.9)).  policies/policy.py:47-53,
# datasets/rl_waymo/dataset.py:361-387, autoregressive_policy.py:73-78.
ZERO_ACTION_TOKEN = 524
ZERO_RTG_BINS = (0, 35, 35)
