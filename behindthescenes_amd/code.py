"""PositionalEncoding with the reference's constructor, buffers and state-dict keys (models/common/model/code.py:6-52).
Inside the renderer the encoding is evaluated by the fused HIP kernel; ``forward`` here exists for API completeness
(stand-alone use on any device) and is not on the render path."""
import numpy as np
import torch


class PositionalEncoding(torch.nn.Module):
    def __init__(self, num_freqs=6, d_in=3, freq_factor=np.pi, include_input=True):
        super().__init__()
        self.num_freqs, self.d_in, self.include_input = num_freqs, d_in, include_input
        self.freq_factor = float(freq_factor)
        self.freqs = freq_factor * 2.0 ** torch.arange(0, num_freqs)
        self.d_out = num_freqs * 2 * d_in + (d_in if include_input else 0)
        self.register_buffer("_freqs", torch.repeat_interleave(self.freqs, 2).view(1, -1, 1))
        phases = torch.zeros(2 * num_freqs)
        phases[1::2] = np.pi * 0.5
        self.register_buffer("_phases", phases.view(1, -1, 1))

    def forward(self, x):
        e = x.unsqueeze(1).repeat(1, self.num_freqs * 2, 1)
        e = torch.sin(torch.addcmul(self._phases, e, self._freqs)).view(x.shape[0], -1)
        return torch.cat((x, e), dim=-1) if self.include_input else e

    @classmethod
    def from_conf(cls, conf, d_in=3):
        return cls(conf.get("num_freqs", 6), d_in, conf.get("freq_factor", np.pi), conf.get("include_input", True))
