from .RFN_new import RFN  # noqa: F401
