"""`Policy` — the plugin base class, same constructor and methods as the reference's policies/policy.py:8-154.

Host-side history buffers are kept exactly as the reference keeps them (float64 NumPy, filled from the per-vehicle
dict of the evaluator); subclasses hand them to the HIP path."""
from __future__ import annotations

import numpy as np


def get_object_type_onehot(agent_type):   # utils/data.py:339-341
    agent_types = {"unset": 0, "vehicle": 1, "pedestrian": 2, "cyclist": 3, "other": 4}
    return np.eye(len(agent_types))[agent_types[agent_type]]


class Policy:
    def __init__(self, cfg, model_path, model, use_rtg, predict_rtgs, discretize_rtgs, real_time_rewards,
                 privileged_return, max_return, min_return, key_dict, tilt_dict, name):
        self.cfg = cfg.copy()
        self.model_path = model_path
        self.model = model
        self.model.eval()
        self.cfg_model = model.cfg.model
        self.cfg_rl_waymo = model.cfg.dataset.waymo
        self.steps = self.cfg.nocturne.steps
        self.use_rtg = use_rtg
        self.predict_rtgs = predict_rtgs
        self.discretize_rtgs = discretize_rtgs
        self.real_time_rewards = real_time_rewards
        self.privileged_return = privileged_return
        self.max_return = max_return
        self.min_return = min_return
        self.key_dict = key_dict
        self.tilt_dict = tilt_dict
        self.name = name

    def reset(self, vehicle_data_dict):
        n = len(vehicle_data_dict.keys())
        self.states = np.zeros((n, self.steps, 8))
        self.gt_states = np.zeros((n, self.steps, 8))
        self.types = np.zeros((n, 5))
        self.actions = np.zeros((n, self.steps, 2))
        self.rtgs = np.zeros((n, self.steps, self.cfg_model.num_reward_components))
        self.goals = np.zeros((n, self.steps, self.cfg_rl_waymo.goal_dim))
        self.timesteps = np.zeros((n, self.steps, 1))
        self.relevant_agent_idxs = {}
        self.idx_to_veh_id = {}
        self.veh_id_to_idx = {}
        for i, v in enumerate(vehicle_data_dict.keys()):
            self.idx_to_veh_id[i] = v
            self.veh_id_to_idx[v] = i

    def update_gt_state(self, gt_data_dict):
        for i, v in enumerate(gt_data_dict.keys()):
            self.gt_states[i] = np.array(gt_data_dict[v]["traj"])[:self.steps]

    def update_state(self, vehicle_data_dict, vehicles_to_evaluate, t):
        for i, v in enumerate(vehicle_data_dict.keys()):
            d = vehicle_data_dict[v]
            self.states[i, t] = (d["position"][t]["x"], d["position"][t]["y"], d["velocity"][t]["x"], d["velocity"][t]["y"],
                                 d["heading"][t], d["length"], d["width"], d["existence"][t])
            if t == 0:
                self.types[i] = get_object_type_onehot(d["type"])
            self.timesteps[i, t] = d["timestep"][t]
            if t > 0:
                self.actions[i, t - 1] = (d["acceleration"][t - 1], d["steering"][t - 1])
                if self.use_rtg:
                    self.rtgs[i, t - 1] = np.array([d[self.key_dict["rtgs"]][t - 1]])[0]
            if self.real_time_rewards and self.use_rtg:
                self.rtgs[i, t] = np.array([d[self.key_dict["rtgs"]][t]])[0]
            gh, gs = d["goal_heading"], d["goal_speed"]
            goal = np.array([d["goal_position"]["x"], d["goal_position"]["y"], gs * np.cos(gh), gs * np.sin(gh), gh])
            self.goals[i, t] = goal[:self.cfg_rl_waymo.goal_dim]

    def get_data(self, gt_data_dict, preproc_data, dset, vehicles_to_evaluate, t):
        pass

    def predict(self, vehicle_data_dict, gt_data_dict, preproc_data, dset, vehicles_to_evaluate, t):
        pass

    def act(self, veh, t, vehicle_data_dict):
        pass
