"""The tiny density MLP with the reference's module/parameter names, so that reference checkpoints load unchanged
(models/common/model/resnetfc.py:10-198, mlp_util.py:5-15).  The modules only OWN the parameters: inside the renderer
they are read by pointer by the fused HIP kernel (``packed()``); the torch ``forward`` is for stand-alone use."""
import torch
from torch import nn


class ResnetBlockFC(nn.Module):
    def __init__(self, size_in, size_out=None, size_h=None, beta=0.0):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        if size_in != size_out or beta > 0:
            raise NotImplementedError("only the size_in == size_out ReLU block used by the shipped configs is supported")
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        nn.init.constant_(self.fc_0.bias, 0.0)
        nn.init.kaiming_normal_(self.fc_0.weight, a=0, mode="fan_in")
        nn.init.constant_(self.fc_1.bias, 0.0)
        nn.init.zeros_(self.fc_1.weight)
        self.activation = nn.ReLU()
        self.shortcut = None

    def forward(self, x):
        return x + self.fc_1(self.activation(self.fc_0(self.activation(x))))


class ResnetFC(nn.Module):
    def __init__(self, d_in, d_out=4, n_blocks=5, d_latent=0, d_hidden=128, beta=0.0, combine_layer=1000,
                 combine_type="average", use_spade=False):
        super().__init__()
        if d_latent != 0 or beta > 0 or use_spade or combine_layer < n_blocks:
            raise NotImplementedError("d_latent / softplus-beta / spade / combine_layer are not used by any shipped config")
        self.lin_in = nn.Linear(d_in, d_hidden)
        nn.init.constant_(self.lin_in.bias, 0.0)
        nn.init.kaiming_normal_(self.lin_in.weight, a=0, mode="fan_in")
        self.lin_out = nn.Linear(d_hidden, d_out)
        nn.init.constant_(self.lin_out.bias, 0.0)
        nn.init.kaiming_normal_(self.lin_out.weight, a=0, mode="fan_in")
        self.n_blocks, self.d_latent, self.d_in, self.d_out, self.d_hidden = n_blocks, d_latent, d_in, d_out, d_hidden
        self.combine_layer, self.combine_type, self.use_spade = combine_layer, combine_type, use_spade
        self.blocks = nn.ModuleList([ResnetBlockFC(d_hidden, beta=beta) for _ in range(n_blocks)])
        self.activation = nn.ReLU()

    def forward(self, zx, combine_inner_dims=(1,), combine_index=None, dim_size=None):
        assert zx.size(-1) == self.d_latent + self.d_in
        x = self.lin_in(zx)
        for blk in self.blocks:
            x = blk(x)
        return self.lin_out(self.activation(x))

    def packed(self) -> torch.Tensor:
        """Flat fp32 parameter vector in the layout of include/bts_render.h (differentiable: autograd splits the gradient of the
        packed vector back onto the individual nn.Parameters).  Cached until a parameter changes (version counter / storage) or
        ``invalidate_packed()`` (BTSNet.encode calls it: one vector -- and one split of its gradient -- per training step, however
        many renders and projections of the step read it; a multiscale step has eight)."""
        params = [self.lin_in.weight, self.lin_in.bias]
        for blk in self.blocks:
            params += [blk.fc_0.weight, blk.fc_0.bias, blk.fc_1.weight, blk.fc_1.bias]
        params += [self.lin_out.weight, self.lin_out.bias]
        key = (torch.is_grad_enabled(),) + tuple((p._version, p.data_ptr(), p.requires_grad) for p in params)
        hit = self.__dict__.get("_packed_cache")
        if hit is not None and hit[0] == key:
            return hit[1]
        out = torch.cat([p.reshape(-1) for p in params])
        self.__dict__["_packed_cache"] = (key, out)
        return out

    def invalidate_packed(self):
        """Drops the cached vector.  ``packed()`` notices in-place edits that bump a parameter's version counter, a new storage and a
        changed ``requires_grad``; edits through ``.data`` (``p.data.copy_()``, an EMA swap, weight clipping) bump nothing -- call this
        after them.  ``load_state_dict`` / ``.to()`` / ``.cuda()`` call it themselves (the hooks below), ``BTSNet.encode`` too."""
        self.__dict__["_packed_cache"] = None

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_packed()
        return super()._apply(fn, *args, **kwargs)

    def __getstate__(self):     # copy.deepcopy / pickle: the cached vector (a non-leaf tensor under autograd) stays behind
        state = self.__dict__.copy()
        state["_packed_cache"] = None
        return state

    @classmethod
    def from_conf(cls, conf, d_in, **kwargs):
        return cls(d_in, n_blocks=conf.get("n_blocks", 5), d_hidden=conf.get("d_hidden", 128), beta=conf.get("beta", 0.0),
                   combine_layer=conf.get("combine_layer", 1000), combine_type=conf.get("combine_type", "average"),
                   use_spade=conf.get("use_spade", False), **kwargs)


def make_mlp(conf, d_in, d_latent=0, allow_empty=False, **kwargs):
    mlp_type = conf.get("type", "mlp")
    if mlp_type == "resnet":
        return ResnetFC.from_conf(conf, d_in, d_latent=d_latent, **kwargs)
    if mlp_type == "empty" and allow_empty:
        return None
    raise NotImplementedError(f"Unsupported MLP type {mlp_type!r} (the shipped configs use 'resnet' / 'empty')")
