from .builder import build_vision_projector  # noqa: F401
