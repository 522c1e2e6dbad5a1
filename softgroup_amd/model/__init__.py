from .softgroup import SoftGroup  # noqa: F401
