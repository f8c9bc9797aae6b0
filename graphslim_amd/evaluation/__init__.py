from .eval_agent import Evaluator  # noqa: F401
