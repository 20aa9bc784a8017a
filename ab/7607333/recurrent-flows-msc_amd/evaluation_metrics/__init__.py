from .error_metrics import Evaluator  # noqa: F401
