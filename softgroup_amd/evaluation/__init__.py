from .instance_eval import ScanNetEval  # noqa: F401
