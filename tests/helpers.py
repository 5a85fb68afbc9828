import os

import numpy as np

from conftest import GOLDEN
from ctrlsim_amd import spec

TINY = dict(dataset__waymo__max_num_agents=4, dataset__waymo__train_context_length=4,
            dataset__waymo__max_num_road_polylines=6, dataset__waymo__max_num_road_pts_per_polyline=8)
LOOP = dict(dataset__waymo__max_num_agents=6, dataset__waymo__train_context_length=8,
            dataset__waymo__max_num_road_polylines=12, dataset__waymo__max_num_road_pts_per_polyline=10,
            nocturne__steps=20)
# SURVEY.md section 8(d), "secondary interpretation (non-reference)": one wide context per scenario instead of ceil(N / 24) focal groups
WIDE = dict(dataset__waymo__max_num_agents=64, dataset__waymo__max_num_road_polylines=512)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def cfg_of(kind, variant=None):
    """variant "il" / "trajeglish": cfgs/model/{il,trajeglish}.yaml on top of the size preset."""
    over = dict({"tiny": TINY, "loop": LOOP, "full": {}, "wide": WIDE}[kind])
    if variant:
        over.update({f"model__{variant}": True, "model__predict_rtg": False, "model__predict_future_states": False})
    return spec.make_cfg(**over)
