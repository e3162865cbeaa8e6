"""Host utilities the entry points use (reference: video_chatgpt/utils.py:92-98)."""


def disable_torch_init():
    """Skip torch's default (re)initialisation of Linear / LayerNorm weights: every parameter is overwritten by
    the checkpoint anyway (video_chatgpt/utils.py:92-98)."""
    import torch
    setattr(torch.nn.Linear, "reset_parameters", lambda self: None)
    setattr(torch.nn.LayerNorm, "reset_parameters", lambda self: None)
