"""Where does the fused step's feature-map gradient differ from the reference's golden (tests/golden/train_step.npz)?  Prints the texels
that carry the error and whether a relu gate of lin_in's output explains them (a gate that differs between two evaluations moves all C
channels of the 4 taps of ONE sample by W_f[u, :] * g_h[u] * bilinear weight)."""
import ast, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bts_oracle as O
from tests.test_gpu_fused_anchor import _net, _fused_step, _oracle_step, _hip_grads, GOLDEN

z = np.load(f"{GOLDEN}/train_step.npz"); meta = ast.literal_eval(str(z["meta"])); t = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
n, pc, ps, K, H, W = meta["n"], meta["patches"], meta["patch"], meta["K"], meta["H"], meta["W"]
cfg = O.FieldConfig(d_min=meta["d_min"], d_max=meta["d_max"])
scene = dict(images=t["images"], feat=t["feat"], projs=t["projs"], poses=t["poses"])
mlp = O.MlpParams(t["w_in"], t["b_in"], [], t["w_out"], t["b_out"])
_, _, truth, _ = _oracle_step(scene, mlp, cfg, meta["ids_render"], t["rays"], t["u"], t["rgb_gt"], K, True, dtype=torch.float64)
ref = t["g_feat"].double()
top = ref.abs().max().item()
for trial in range(3):
    net = _net(cfg, mlp, scene, H, W, meta["C"], train=True)
    step = _fused_step(net, K, pc * ps * ps, cfg, hard_cap=True)
    step.fused = trial != 2
    torch.manual_seed(meta["seed"] + 1)
    kw = dict(jitter=t["u"].cuda()) if step.fused else {}
    if not step.fused:
        print("(entry-by-entry path draws its own jitter: skipped)"); break
    loss, parts, data = step(t["images"].cuda(), t["projs"].cuda(), t["poses"].cuda(), ids_encoder=[0], ids_render=meta["ids_render"], ids_loss=meta["ids_loss"], **kw)
    loss.backward()
    g = _hip_grads(net)["feat"].double().view_as(ref)
    e = (g - ref).abs() / top
    per_texel = e.amax(dim=1)          # (n, H, W)
    idx = per_texel.flatten().argsort(descending=True)[:12]
    print(f"trial {trial}: max {e.max().item():.2e}, L2 {((g-ref).norm()/ref.norm()).item():.2e}; texels with err > 2e-5: {(per_texel > 2e-5).sum().item()} of {per_texel.numel()}")
    for i in idx.tolist():
        b, y, x = i // (H * W), (i // W) % H, i % W
        ch = e[b, :, y, x]
        print(f"   texel (n={b}, y={y}, x={x}): max {ch.max().item():.2e}, channels > 1e-5: {(ch > 1e-5).sum().item()} / {ch.numel()}, |g| there {ref[b,:,y,x].abs().max().item()/top:.2e}")
    # without the worst 8 texels
    mask = torch.ones_like(per_texel, dtype=torch.bool).flatten(); mask[idx[:8]] = False
    m = mask.view_as(per_texel).unsqueeze(1).expand_as(e)
    print(f"   without the 8 worst texels: max {e[m].max().item():.2e}, L2 {((g-ref)[m].norm()/ref.norm()).item():.2e}")
    gw = _hip_grads(net)["lin_in.weight"].double(); rw = t["g_w_in"].double()
    ew = (gw - rw).abs() / rw.abs().max()
    rows = ew.amax(dim=1)
    print("   lin_in.weight rows (hidden units) by error:", [(int(i), f"{rows[i].item():.1e}") for i in rows.argsort(descending=True)[:5].tolist()])
