from .policy_evaluator import PolicyEvaluator  # noqa: F401
from .planner_adversary_evaluator import PlannerAdversaryEvaluator  # noqa: F401
