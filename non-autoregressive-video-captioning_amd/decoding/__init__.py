from .na_generate import generate  # noqa: F401
