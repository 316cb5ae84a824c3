#!/usr/bin/env python3
"""Known-answer vector for TractOracle-Net: the
REFERENCE's TransformerOracle (TrackToLearn/oracles/transformer_oracle.py,
importable as-is) with seeded random weights, CPU float32, eval mode.  The
trained checkpoint is not shipped, so n_head / n_layers here are
arbitrary small values."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from make_golden import _save  # noqa: E402

sys.path.insert(0, '/root/reference')
from TrackToLearn.oracles.transformer_oracle import TransformerOracle  # noqa: E402

torch.manual_seed(3)
model = TransformerOracle(input_size=381, output_size=1, n_head=4, n_layers=2,
                          lr=1e-4)
model.eval()
# the two big feed-forward matrices per layer are rounded to float16 values
# first, so that the fixture can store them losslessly at half the size
with torch.no_grad():
    for name, prm in model.named_parameters():
        if 'linear' in name and prm.ndim == 2:
            prm.copy_(prm.half().float())
rng = np.random.RandomState(1)
x = (rng.standard_normal((24, 127, 3)) * 0.5).astype(np.float32)
with torch.no_grad():
    y = model(torch.from_numpy(x)).numpy()
out = {'x': x, 'y': y, 'input_size': 381, 'n_head': 4, 'n_layers': 2}
for k, v in model.state_dict().items():
    out['sd/' + k] = v.numpy().astype(np.float16 if 'linear' in k and v.ndim == 2
                                      else np.float32)
_save('oracle_transformer', out)
