from . import Constants  # noqa: F401
