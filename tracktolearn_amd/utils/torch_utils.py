"""Device helpers (TrackToLearn/utils/torch_utils.py:3-15).  On this stack
"cuda" is the MI355X through PyTorch-ROCm."""
import torch


def get_device():
    if torch.cuda.is_available():
        return torch.device('cuda')
    return torch.device('cpu')


def assert_accelerator():
    assert torch.cuda.is_available(), 'an MI355X (torch "cuda" device) is required'


def get_device_str():
    return str(get_device())
