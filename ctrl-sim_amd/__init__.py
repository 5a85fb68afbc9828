"""MI355X-native closed-loop rollout path of CtRL-Sim.

The directory is named `ctrl-sim_amd` (repo contract); import it as `ctrlsim_amd`
(the top-level `ctrlsim_amd.py` aliases this directory as a package).
"""
from .spec import make_cfg, Dims, Cfg  # noqa: F401

__all__ = ["make_cfg", "Dims", "Cfg"]
