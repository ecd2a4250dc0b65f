from .select_backbone import select_backbone  # noqa: F401
