from .synthetic import SyntheticMovingMNIST  # noqa: F401
