"""TractOracle-Net as one HIP kernel (``ttl_oracle_net_forward``,
csrc/ttl_oracle_net.hip): host side.

``pack_oracle_net`` lays the weights of a ``TransformerOracle``
(TrackToLearn/oracles/transformer_oracle.py:38-118) out in the order the
kernel's MFMA operands want them -- fp16, 32x32 blocks as fragments of
``v_mfma_f32_32x32x16_f16`` with the k index permuted the way an accumulator
tile presents its rows (element j of lane half h of k-step s is k = 16 s +
8 (j >> 2) + 4 h + (j & 3)), per-row vectors (biases, LayerNorm gains) in the
accumulator's row order -- once per model; ``FusedOracleNet.__call__`` then
scores a batch of segment-vector sequences in one launch.

Supported: the reference's architecture as it is instantiated (d_model 32,
128 tokens = 127 segments + CLS, ReLU post-norm encoder layers, no final
norm, feed-forward width a multiple of 32 up to 8 192) with 1, 2 or 4 heads.  Anything
else keeps the PyTorch-ROCm module (``OracleSingleton`` checks
``FusedOracleNet.supports``).  Arithmetic follows ``torch.autocast(fp16)``,
what the reference runs the network under (oracles/oracle.py:76): fp16
operands, fp32 accumulation, Linear outputs rounded to fp16, softmax and
LayerNorm in fp32.
"""
import ctypes as C

import torch
from torch import nn

from tracktolearn_amd import _lib

D_MODEL, TOKENS = 32, 128


def _kperm():
    """k index of element j of lane half h of k-step s: [2 s][2 h][8 j]."""
    s = torch.arange(2).view(2, 1, 1)
    h = torch.arange(2).view(1, 2, 1)
    j = torch.arange(8).view(1, 1, 8)
    return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)


def _fperm():
    """row of accumulator register a of lane half h: [2 h][16 a]."""
    h = torch.arange(2).view(2, 1)
    a = torch.arange(16).view(1, 16)
    return 8 * (a >> 2) + 4 * h + (a & 3)


def pack32(w):
    """A [32 x 32] block -> [2 s][64 lanes][8] fragment elements: lane (r =
    lane & 31, h = lane >> 5) holds w[r][kperm(s, h, j)] (the A operand of
    w . X, or the B operand of X . w^T)."""
    assert tuple(w.shape) == (32, 32)
    kp = _kperm()                                        # [s][h][j]
    lane = torch.arange(64)
    r, h = lane & 31, lane >> 5
    cols = kp[:, h, :]                                   # [s][64][8]
    return w[r.view(1, 64, 1).expand(2, 64, 8), cols]


def rowpack(v):
    """A [32] per-row vector -> [2 h][16 a] in accumulator row order."""
    return v[_fperm()]


def pack_oracle_net(model, device):
    """dict of device tensors for ``ttl_oracle_net_forward``."""
    layers = list(model.bert.layers)
    ff = layers[0].linear1.out_features
    chunks = ff // 32
    halves, floats = [], []
    for layer in layers:
        at = layer.self_attn
        w_in, b_in = at.in_proj_weight.detach().float().cpu(), at.in_proj_bias.detach().float().cpu()
        w_o, b_o = at.out_proj.weight.detach().float().cpu(), at.out_proj.bias.detach().float().cpu()
        w1, b1 = layer.linear1.weight.detach().float().cpu(), layer.linear1.bias.detach().float().cpu()
        w2, b2 = layer.linear2.weight.detach().float().cpu(), layer.linear2.bias.detach().float().cpu()
        frag = [pack32(w_in[0:32]), pack32(w_in[32:64]), pack32(w_in[64:96]), pack32(w_o)]
        frag += [pack32(w1[32 * c:32 * c + 32]) for c in range(chunks)]
        frag += [pack32(w2[:, 32 * c:32 * c + 32]) for c in range(chunks)]
        halves.append(torch.stack(frag).reshape(-1))     # [(4 + 2C)][2][64][8]
        vec = [rowpack(b_in[0:32]), rowpack(b_in[32:64]), b_in[64:96].view(2, 16),
               rowpack(b_o),
               rowpack(layer.norm1.weight.detach().float().cpu()),
               rowpack(layer.norm1.bias.detach().float().cpu()),
               rowpack(b2),
               rowpack(layer.norm2.weight.detach().float().cpu()),
               rowpack(layer.norm2.bias.detach().float().cpu())]
        vec += [rowpack(b1[32 * c:32 * c + 32]) for c in range(chunks)]
        floats.append(torch.stack(vec).reshape(-1))      # 288 + 32 C
    emb = model.embedding[0]
    we, be = emb.weight.detach().float().cpu(), emb.bias.detach().float().cpu()
    # fp16-rounded, as autocast feeds them to the Linear
    e4 = torch.cat([we.half().float(), be.half().float().view(32, 1)], dim=1)       # [32][4]
    pe = model.pos_encoding.pe[:TOKENS, 0].detach().float().cpu()                   # [128][32]
    lane = torch.arange(64)
    n, h = lane & 31, lane >> 5
    fp = _fperm()                                                                   # [h][a]
    pe_pack = torch.stack([pe[(32 * nt + n).view(64, 1).expand(64, 16), fp[h]]
                           for nt in range(4)])                                      # [4][64][16]
    head = torch.cat([rowpack(model.head.weight.detach().float().cpu()[0]).reshape(-1),
                      model.head.bias.detach().float().cpu().view(1)])
    dev = torch.device(device)
    return {
        'wh': torch.stack(halves).half().contiguous().to(dev),
        'wf': torch.stack(floats).contiguous().to(dev),
        'embed': e4[_fperm()].contiguous().to(dev),        # [2][16][4]
        'cls': model.cls_token.detach().float().contiguous().to(dev),
        'pe': pe_pack.contiguous().to(dev),
        'head': head.contiguous().to(dev),
        'n_layers': len(layers), 'n_head': layers[0].self_attn.num_heads, 'ff': ff,
    }


class FusedOracleNet:
    """``scores = FusedOracleNet(model)(dirs)`` for dirs (N, 127, 3) float32 on
    the model's CUDA device: (N,) float32 scores in (0, 1)."""

    @staticmethod
    def supports(model):
        """Whether ``model`` is the architecture the kernel implements."""
        try:
            layers = list(model.bert.layers)
            l0 = layers[0]
            return (model.embedding_size == D_MODEL and model.input_size // 3 + 1 == TOKENS
                    and model.output_size == 1 and model.bert.norm is None and len(layers) >= 1
                    and all(isinstance(m, nn.TransformerEncoderLayer) for m in layers)
                    and l0.self_attn.num_heads in (1, 2, 4) and l0.self_attn.batch_first
                    and not l0.norm_first and l0.linear1.out_features % 32 == 0
                    and l0.linear1.out_features <= 8192
                    and getattr(l0.activation, '__name__', '') == 'relu'
                    and l0.self_attn.in_proj_weight is not None
                    and abs(l0.norm1.eps - 1e-5) < 1e-12
                    and isinstance(model.embedding[0], nn.Linear)
                    and model.embedding[0].in_features == 3)
        except (AttributeError, IndexError, TypeError):
            return False

    def __init__(self, model, device=None):
        if not self.supports(model):
            raise ValueError('FusedOracleNet: unsupported architecture')
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None
                                   else next(model.parameters()).device)
        if self.device.type != 'cuda':
            raise _lib.TTLError('FusedOracleNet needs a CUDA device: there is no CPU path')
        self.p = pack_oracle_net(model, self.device)

    def __call__(self, dirs):
        n = dirs.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        if tuple(dirs.shape[1:]) != (TOKENS - 1, 3):
            raise ValueError(f'FusedOracleNet: expected (N, {TOKENS - 1}, 3) segment vectors')
        dirs = dirs.to(self.device, torch.float32).contiguous()
        p = self.p
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ttl_oracle_net_forward(
                dirs.data_ptr(), n, p['wh'].data_ptr(), p['wf'].data_ptr(), p['embed'].data_ptr(),
                p['cls'].data_ptr(), p['pe'].data_ptr(), p['head'].data_ptr(), p['n_layers'],
                p['n_head'], p['ff'], out.data_ptr(), stream), 'ttl_oracle_net_forward')
        return out
