from .policy_evaluator import PolicyEvaluator  # noqa: F401
