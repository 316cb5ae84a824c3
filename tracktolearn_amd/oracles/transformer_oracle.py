"""TractOracle-Net: the transformer that scores streamlines.

Same architecture, parameter names and checkpoint format as
TrackToLearn/oracles/transformer_oracle.py (embedding Linear(3, 32) + ReLU,
learned 3-vector CLS token prepended to the sequence, sinusoidal positional
encoding, ``nn.TransformerEncoder`` of ``n_layers`` post-norm layers with
``n_head`` heads and the default 2048-wide feed-forward, linear head on the
CLS position, sigmoid), running on PyTorch-ROCm.
"""
import math

import torch
from torch import nn


class PositionalEncoding(nn.Module):
    """Fixed sinusoidal table added to a batch-first sequence
    (transformer_oracle.py:7-35)."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) *
                             (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term)
        pe[:, 0, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe)

    def forward(self, x):
        # x: (batch, seq, d); the table is (seq, 1, d)
        return self.dropout(x + self.pe[:x.size(1)].transpose(0, 1))


class TransformerOracle(nn.Module):
    """transformer_oracle.py:38-118."""

    def __init__(self, input_size, output_size, n_head, n_layers, lr):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.lr = lr
        self.n_head = n_head
        self.n_layers = n_layers
        self.embedding_size = 32
        self.cls_token = nn.Parameter(torch.randn((3)))
        layer = nn.TransformerEncoderLayer(self.embedding_size, n_head,
                                           batch_first=True)
        self.embedding = nn.Sequential(nn.Linear(3, self.embedding_size),
                                       nn.ReLU())
        self.pos_encoding = PositionalEncoding(
            self.embedding_size, max_len=(input_size // 3) + 1)
        self.bert = nn.TransformerEncoder(layer, self.n_layers)
        self.head = nn.Linear(self.embedding_size, output_size)
        self.sig = nn.Sigmoid()

    def forward(self, x):
        """x: (N, L, 3) segment vectors -> (N,) scores in (0, 1)."""
        n = x.shape[0]
        cls_tokens = self.cls_token.to(x.dtype).expand(n, 1, 3)
        x = torch.cat((cls_tokens, x), dim=1)
        x = self.embedding(x) * math.sqrt(self.embedding_size)
        hidden = self.bert(self.pos_encoding(x))
        return self.sig(self.head(hidden[:, 0])).squeeze(-1)

    @classmethod
    def load_from_checkpoint(cls, checkpoint: dict):
        hp = checkpoint['hyper_parameters']
        model = cls(hp['input_size'], hp['output_size'], hp['n_head'],
                    hp['n_layers'], hp['lr'])
        model.load_state_dict(checkpoint['state_dict'])
        model.eval()
        return model


def save_random_checkpoint(path, n_head=4, n_layers=4, input_size=381, seed=0):
    """A seeded random-init checkpoint in the reference's format -- the
    trained ``tractoracle.ckpt`` is not distributed with the reference
    (SURVEY F11), so synthetic runs (BASELINE config 5) use this."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = TransformerOracle(input_size, 1, n_head, n_layers, 1e-4)
    torch.random.set_rng_state(gen_state)
    torch.save({'hyper_parameters': {'name': 'TransformerOracle',
                                     'input_size': input_size,
                                     'output_size': 1, 'n_head': n_head,
                                     'n_layers': n_layers, 'lr': 1e-4},
                'state_dict': model.state_dict()}, path)
    return path
