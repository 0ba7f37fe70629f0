from .dual_form import ttt_linear, ttt_mlp  # noqa: F401
