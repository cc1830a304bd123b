import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, copy
from test_product_model_gpu import _setup, _rel
from dfine_seg_amd.layers import ConvNorm
g, model, crit, x, targets, inject, seg = _setup("model_m640")
model.train()
names = [(n, m) for n, m in model.backbone.named_modules() if isinstance(m, ConvNorm)][:14]
def run(ac):
    outs = {}
    hs = []
    for n, m in names:
        hs.append(m.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n, o.detach().float().cpu())))
        hs.append(m.conv.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n + ".conv", o.detach().float().cpu())))
        bn = getattr(m, m.norm_name)
        hs.append(bn.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n + ".bn", o.detach().float().cpu())))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
        model.backbone(x.contiguous(memory_format=torch.channels_last))
    for h in hs: h.remove()
    return outs
state = copy.deepcopy(model.state_dict())
a = run(False); model.load_state_dict(state); b = run(True)
for n, _ in names:
    print(f"{n:45s} conv {_rel(b[n+'.conv'], a[n+'.conv']):.4f}  bn {_rel(b[n+'.bn'], a[n+'.bn']):.4f}  out {_rel(b[n], a[n]):.4f}")
