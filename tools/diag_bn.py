import torch
torch.manual_seed(0)
def rel(a,b): return float((a.float()-b.float()).norm()/b.float().norm())
for mean,std in [(0,1),(3,1),(10,1),(10,0.1)]:
    x=(torch.randn(2,32,160,160,device="cuda")*std+mean).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bn=torch.nn.BatchNorm2d(32).cuda()
    ref=bn(x.float())
    for en in (True, False):
        torch.backends.cudnn.enabled=en
        y=bn(x)
        print(f"mean {mean} std {std} miopen={en}: bn(bf16) vs bn(fp32 of same input) rel {rel(y,ref):.4f} out dtype {y.dtype}")
    torch.backends.cudnn.enabled=True
