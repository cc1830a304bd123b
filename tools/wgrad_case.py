"""One wgrad shape, a few launches -- the subject of rocprofv3 --pmc runs.
    python tools/wgrad_case.py conv B H CI CO k     |     python tools/wgrad_case.py linear M N K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfine_seg_amd import _lib
lib = _lib.load()
kind = sys.argv[1]
if kind == "conv":
    B, H, ci, co, k = map(int, sys.argv[2:7])
    x = torch.randn(B, H, H, ci, device="cuda", dtype=torch.bfloat16)
    gy = torch.randn(B, H, H, co, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(co, k, k, ci, device="cuda")
    run = lambda: lib.dfine_conv_wgrad(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(dw), B, H, H, ci, H, H, co, k, k, 1, k // 2, co, ci, _lib.stream_ptr())
else:
    M, N, K = map(int, sys.argv[2:5])
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    gy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(N, K, device="cuda")
    run = lambda: lib.dfine_linear_wgrad(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(dw), None, M, N, K, K, N, K, _lib.stream_ptr())
for _ in range(5):
    assert run() == 0
torch.cuda.synchronize()
