import sys, collections, traceback; sys.path.insert(0, "/root/repo")
import torch
import dfine_seg_amd
from dfine_seg_amd import ops, gemm
from dfine_seg_amd.synth import synthetic_batch
from dfine_seg_amd.train import TrainStep
log = collections.Counter(); nbytes = collections.Counter()
orig_contig = torch.Tensor.contiguous
def contig(self, *a, **k):
    mf = k.get("memory_format", a[0] if a else torch.contiguous_format)
    if self.is_cuda and self.numel() > 100000 and not self.is_contiguous(memory_format=mf):
        st = traceback.extract_stack(limit=6)
        site = " <- ".join(f"{f.name}:{f.lineno}" for f in st[-4:-1])
        log[(site, tuple(self.shape))] += 1; nbytes[(site, tuple(self.shape))] += self.numel() * self.element_size()
    return orig_contig(self, *a, **k)
torch.Tensor.contiguous = contig
t = TrainStep("m", 80, False, (640, 640), "cuda", total_steps=1000, use_graph=False)
images, targets = synthetic_batch(32, 640, 640, 80, 8, seed=42, device="cuda")
images = images.contiguous(memory_format=torch.channels_last)
for _ in range(2): t.step(images, targets)
log.clear(); nbytes.clear()
t.step(images, targets); torch.cuda.synchronize()
tot = 0
for k, v in sorted(nbytes.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{v/1e6:8.1f} MB x{log[k]:3d} {k[1]} {k[0]}"); 
print("total MB", sum(nbytes.values())/1e6)
