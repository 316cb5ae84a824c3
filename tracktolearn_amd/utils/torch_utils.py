"""Device selection.  On this stack the torch "cuda" device IS the MI355X
(PyTorch-ROCm); there is no MPS / CPU training path
(cf. TrackToLearn/utils/torch_utils.py)."""
import torch


def get_device():
    return torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def get_device_str():
    return get_device().type


def assert_accelerator():
    if not torch.cuda.is_available():
        raise AssertionError('an MI355X (torch "cuda" device) is required')
