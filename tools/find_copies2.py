"""Python-level layout copies of one eager step (forward and the custom Functions' backward): which
torch.cat / .contiguous() / F.pad / .clone() calls move how many bytes, by call site."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dfine_seg_amd.synth import synthetic_batch
from dfine_seg_amd.train import TrainStep

t = TrainStep("m", 80, False, (640, 640), "cuda", total_steps=1000, use_graph=False)
images, targets = synthetic_batch(32, 640, 640, 80, 8, seed=42, device="cuda")
images = images.contiguous(memory_format=torch.channels_last)
for _ in range(3):
    t.step(images, targets)
log = collections.defaultdict(lambda: [0, 0])

def site():
    frames = [fr for fr in traceback.extract_stack()[:-2] if "d-fine-seg_amd" in fr.filename]
    if not frames:
        return "?"
    fr = frames[-1]
    if fr.name == "_nhwc" and len(frames) > 1:  # the helper: report its caller
        fr = frames[-2]
    return f"{os.path.basename(fr.filename)}:{fr.lineno} [{fr.name}] {fr.line[:50] if fr.line else ''}"

orig_cat, orig_contig, orig_pad, orig_clone = torch.cat, torch.Tensor.contiguous, F.pad, torch.Tensor.clone
def cat(ts, *a, **k):
    out = orig_cat(ts, *a, **k)
    e = log[("cat", site())]; e[0] += out.numel() * out.element_size(); e[1] += 1
    return out
def contig(self, *a, **k):
    out = orig_contig(self, *a, **k)
    if out.data_ptr() != self.data_ptr():
        e = log[("contiguous", site() + f" {tuple(self.shape)}")]; e[0] += out.numel() * out.element_size(); e[1] += 1
    return out
def pad(x, *a, **k):
    out = orig_pad(x, *a, **k)
    e = log[("pad", site())]; e[0] += out.numel() * out.element_size(); e[1] += 1
    return out
def clone(self, *a, **k):
    out = orig_clone(self, *a, **k)
    e = log[("clone", site())]; e[0] += out.numel() * out.element_size(); e[1] += 1
    return out
torch.cat, torch.Tensor.contiguous, F.pad, torch.Tensor.clone = cat, contig, pad, clone
t.step(images, targets)
torch.cuda.synchronize()
torch.cat, torch.Tensor.contiguous, F.pad, torch.Tensor.clone = orig_cat, orig_contig, orig_pad, orig_clone
tot = sum(v[0] for v in log.values())
print(f"total {tot/1e6:.1f} MB written by python-level copies")
for (kind, where), (b, n) in sorted(log.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{b/1e6:9.1f} MB n={n:3d} {kind:10s} {where}")
