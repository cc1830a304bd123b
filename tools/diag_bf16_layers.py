"""Where does bf16 round-off grow? Per ConvNorm output of backbone + encoder (D-FINE-m, name-hashed weights, the 2-image
golden batch): relative L2 distance to the fp32 run, for (a) the product's bf16 path (HIP kernels) and (b) stock PyTorch
bf16 autocast on the same modules (F.conv2d / nn.BatchNorm2d / F.relu / LAB = the reference's AMP pipeline on this GPU).
Also prints, per layer, the smallest per-channel batch std relative to the channel's |mean| (BatchNorm divides by it)."""
import copy, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import torch.nn.functional as F
from test_product_model_gpu import _setup, _rel
from dfine_seg_amd import ops
from dfine_seg_amd.layers import ConvNorm

g, model, crit, x, targets, inject, seg = _setup("model_m640")
model.train()
mods = [(n, m) for n, m in model.named_modules() if isinstance(m, ConvNorm) and not n.startswith("decoder")]
state = copy.deepcopy(model.state_dict())

def run(autocast):
    model.load_state_dict(state)
    outs, hs = {}, []
    for n, m in mods:
        hs.append(m.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n, o.detach().float().cpu())))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        model.encoder(model.backbone(x.contiguous(memory_format=torch.channels_last)))
    for h in hs: h.remove()
    return outs

ref = run(False)
mine = run(True)
# stock path: every fused op declines, convs are F.conv2d
saved = (ops.batch_norm_act_supported, ops.dense_conv, ops.depthwise_conv2d, ops.stem_pool_supported, ops.add_layer_norm_supported,
         ops.linear, ops.self_attention_supported)
ops.batch_norm_act_supported = lambda x, c: False
ops.dense_conv = lambda x, w, stride=1, padding=0, pad_br=None, stats=None, out=None: F.conv2d(F.pad(x, (0, pad_br, 0, pad_br)) if pad_br else x, w, None, stride, padding)
ops.depthwise_conv2d = lambda x, w, stride=1: F.conv2d(x, w, None, stride, (w.shape[-1] - 1) // 2, 1, x.shape[1])
ops.stem_pool_supported = lambda x: False
ops.add_layer_norm_supported = lambda x, d: False
ops.linear = lambda x, w, b=None, r0=0, r1=None, relu=False: (F.relu if relu else (lambda t: t))(F.linear(x, w[r0:r1], b[r0:r1] if b is not None else None))
ops.self_attention_supported = lambda qk, v, h: False
stock = run(True)
(ops.batch_norm_act_supported, ops.dense_conv, ops.depthwise_conv2d, ops.stem_pool_supported, ops.add_layer_norm_supported, ops.linear,
 ops.self_attention_supported) = saved
print(f"{'layer':58s} {'product bf16':>12s} {'stock bf16':>12s}")
for n, _ in mods:
    print(f"{n:58s} {_rel(mine[n], ref[n]):12.4f} {_rel(stock[n], ref[n]):12.4f}")
