from .ctrl_sim import CtRLSim  # noqa: F401
