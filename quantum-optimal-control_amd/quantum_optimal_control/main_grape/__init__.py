from .grape import Grape  # noqa: F401
