"""ttt_amd - host-side mirror (Python 3.10, PyTorch-ROCm) of the part of the reference's
``ttt`` package that sits on the TTT-layer + DiT-block hot path, with the
TTT scan running on hand-written gfx950 HIP kernels through ``test_time_training``.

Module/class names, constructor arguments, state-dict keys and tensor contracts are those of the
reference (``ttt.models.ssm.ttt_layer``, ``ttt.models.ssm.mlp_tk``, ``ttt.models.cogvideo.dit`` ...)
so that reference checkpoints load and its drivers can target these classes.
"""
