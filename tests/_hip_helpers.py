"""Builds the drop-in modules (behindthescenes_amd.BTSNet / NeRFRenderer) from a golden case or a synthetic scene."""
import torch

import behindthescenes_amd as bts
from oracle import bts_oracle as O


def make_conf(cfg: O.FieldConfig, C, Hd, nb, H, W):
    return dict(z_near=cfg.d_min, z_far=cfg.d_max, inv_z=cfg.inv_z, learn_empty=cfg.learn_empty, empty_empty=cfg.empty_empty,
                code_mode=cfg.code_mode,
                code=dict(num_freqs=cfg.num_freqs, freq_factor=cfg.freq_factor, include_input=cfg.include_input),
                encoder=dict(type="feature_map", size=(H, W), d_out=C),
                mlp_coarse=dict(type="resnet", n_blocks=nb, d_hidden=Hd), mlp_fine=dict(type="empty"))


def load_mlp(net, mlp: O.MlpParams):
    with torch.no_grad():
        m = net.mlp_coarse
        m.lin_in.weight.copy_(mlp.w_in), m.lin_in.bias.copy_(mlp.b_in)
        for blk, (w0, b0, w1, b1) in zip(m.blocks, mlp.blocks):
            blk.fc_0.weight.copy_(w0), blk.fc_0.bias.copy_(b0), blk.fc_1.weight.copy_(w1), blk.fc_1.bias.copy_(b1)
        m.lin_out.weight.copy_(mlp.w_out), m.lin_out.bias.copy_(mlp.b_out)


def build_net(cfg, mlp, scene, ids_render, empty_feature=None, device="cuda", train=False):
    n, C, H, W = scene["feat"].shape
    net = bts.BTSNet(make_conf(cfg, C, mlp.w_in.shape[0], len(mlp.blocks), H, W))
    load_mlp(net, mlp)
    with torch.no_grad():
        net.encoder.feats[0].data = scene["feat"].clone()
        if empty_feature is not None:
            net.empty_feature.copy_(empty_feature)
    net = net.to(device)
    net.train(train)
    net.encode(scene["images"].to(device), scene["projs"].to(device), scene["poses"].to(device), ids_encoder=[0],
               ids_render=list(ids_render))
    return net


def net_from_case(case, device="cuda", train=False):
    return build_net(case.cfg, case.mlp, case.scene, case.meta["ids_render"], case.t.get("empty_feature"), device, train)


def err_stats(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    d = (a - b).abs()
    return d.max().item(), (d / b.abs().clamp_min(1e-12)).max().item()
