"""Host side of the implicit-GEMM kernel (csrc/igemm.hip, ``dfine_igemm_bf16``): builds the tap
geometry for convolution forward / input gradient (any stride, asymmetric padding), the 2x2 stride-2
transposed convolution of the mask head, and plain Linear GEMMs, on channel-last bf16 tensors that may
be channel SLICES of a wider map (row pitch ``lda`` / ``ldc``), so concat buffers are read and written
in place.

Everything here works on "pixel rows": a tensor [B, H, W, C] (or [M, C]) whose last dimension is
contiguous and whose pixels are ``ld`` elements apart.
"""
import ctypes

import torch

from . import _lib


def pixel_pitch(t):
    """Row pitch (elements) of a [B, H, W, C] / [M, C] tensor whose pixels are equally spaced rows, or None."""
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        return None
    ld = t.stride(-2) if t.dim() >= 2 else t.shape[-1]
    if t.dim() == 2:
        return ld if ld >= t.shape[1] else None
    if t.dim() != 4:
        return None
    B, H, W, C = t.shape
    if ld < C:
        return None
    ok = (W == 1 or t.stride(2) == ld) and (H == 1 or t.stride(1) == W * ld) and (B == 1 or t.stride(0) == H * W * ld)
    return ld if ok else None


def as_pixel_rows(t):
    """(tensor, pitch): ``t`` itself when it already is a pixel-row view, else a dense copy."""
    ld = pixel_pitch(t)
    if ld is None or t.dtype != torch.bfloat16:
        t = t.to(torch.bfloat16).contiguous()
        ld = t.shape[-1]
    return t, ld


def nhwc(t):
    """[B, C, H, W] (channels_last memory, possibly a channel slice) -> its [B, H, W, C] view."""
    return t.permute(0, 2, 3, 1)


def _geom(**kw):
    g = _lib.IGemmGeom()
    taps = kw.pop("taps", [])
    for k, v in kw.items():
        setattr(g, k, int(v))
    if len(taps) > _lib.IGEMM_MAX_TAPS:
        raise _lib.DfineHipError(f"igemm: {len(taps)} taps > {_lib.IGEMM_MAX_TAPS}")
    g.ntaps = len(taps)
    for i, (dy, dx, w) in enumerate(taps):
        g.tap_dy[i], g.tap_dx[i], g.tap_w[i] = dy, dx, w
    return g


import os as _os

_NO_PUBLISH = bool(_os.environ.get("DFINE_IGEMM_STATS_NOPUBLISH"))  # diagnostic (tools/bench_igemm_cold.py)


class BnRed:
    """What an input-gradient launch needs to take the PRODUCING norm's backward sums in its epilogue (include/dfine_hip.h:
    dfine_igemm_bf16_bnred): the norm's input x [n_pix, C] bf16 pixel rows, the optional second gradient part (a channel
    slice of a concat buffer's gradient), saved mean / rstd, gamma / beta, LAB scale, activation code and the zeroed fp64
    reduction workspace (as its fp32 view)."""

    __slots__ = ("x", "dy2", "mean", "rstd", "gamma", "beta", "lab_scale", "red", "act")

    def __init__(self, x, dy2, mean, rstd, gamma, beta, lab_scale, red, act):
        self.x, self.dy2, self.mean, self.rstd, self.gamma, self.beta = x, dy2, mean, rstd, gamma, beta
        self.lab_scale, self.red, self.act = lab_scale, red, act


def _launch(A, Bw, C, bias, stats, g, name, detail, bnred=None):
    """``bnred`` (a BnRed): try the launch with the norm reductions in its epilogue; returns True when that form ran, False
    when the plain launch ran (shapes off the 16-byte paths)."""
    lib = _lib.load()
    if _NO_PUBLISH:
        g.act |= int(_os.environ["DFINE_IGEMM_STATS_NOPUBLISH"]) << 8
    if _os.environ.get("DFINE_IGEMM_DEBUG"):  # diagnostics: 8 = no MFMA, 16 = no DMA (LDS-DMA body, tools/bench_igemm_cfg.py), 32 = bnred epilogue without its atomics
        g.act |= int(_os.environ["DFINE_IGEMM_DEBUG"]) << 8
    _lib.require_gpu(A, Bw, C)
    # timer names say which kernel body serves the launch (csrc/igemm.hip dispatch): plain rows on the 16-byte path -> the
    # LDS-DMA ring (igemm2_kernel, "<site>_rows"), everything else -> the register-staged body (igemm_kernel)
    if not g.gather and g.K % 8 == 0 and g.lda % 8 == 0 and g.ldb % 8 == 0 and A.data_ptr() % 16 == 0 and Bw.data_ptr() % 16 == 0:
        name += "_rows"
    nbytes = (g.M * (g.CI if g.gather else g.K) + g.M * g.N + g.N * g.K) * 2
    if bnred is not None and bias is None and stats is None:
        ldx = pixel_pitch(bnred.x)
        ld2 = pixel_pitch(bnred.dy2) if bnred.dy2 is not None else 0
        if ldx is not None and ld2 is not None:
            br = _lib.IGemmBnRed(x=bnred.x.data_ptr(), dy2=bnred.dy2.data_ptr() if bnred.dy2 is not None else None,
                                 mean=bnred.mean.data_ptr(), rstd=bnred.rstd.data_ptr(), gamma=bnred.gamma.data_ptr(),
                                 beta=bnred.beta.data_ptr(), lab_scale=bnred.lab_scale.data_ptr() if bnred.lab_scale is not None else None,
                                 red=bnred.red.data_ptr(), ldx=ldx, ld_dy2=ld2, act=bnred.act)
            extra = (g.M * g.N * (2 if bnred.dy2 is not None else 1)) * 2
            with _lib.timed(name, nbytes + extra, detail + " +bnred", flops=2.0 * g.M * g.N * g.K):
                rc = lib.dfine_igemm_bf16_bnred(_lib.ptr(A), _lib.ptr(Bw), _lib.ptr(C), ctypes.byref(g), ctypes.byref(br), _lib.stream_ptr())
            if rc == 0:
                return True
            if rc != -2:
                _lib.check(rc, "dfine_igemm_bf16_bnred")
    with _lib.timed(name, nbytes, detail, flops=2.0 * g.M * g.N * g.K):
        rc = lib.dfine_igemm_bf16(_lib.ptr(A), _lib.ptr(Bw), _lib.ptr(C), _lib.ptr(bias), _lib.ptr(stats), ctypes.byref(g),
                                  _lib.stream_ptr())
    _lib.check(rc, "dfine_igemm_bf16")
    return False


def _check_bf16(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.bfloat16:
            raise _lib.DfineHipError(f"igemm operands must be bfloat16, got {t.dtype}")


def linear_forward(x, w, bias=None, relu=False, out=None, name="gemm_fwd", relu_mask=None):
    """x [M, K] rows (pitch >= K), w [N, K] dense bf16, bias f32 [N] -> [M, N] bf16 (``out``: a pixel-row view).
    ``relu_mask`` ([M, N] bf16 rows; input-gradient launches): the result is zeroed where ``relu_mask <= 0`` in the GEMM's
    epilogue (dfine_igemm_bf16_relumask) -- the backward of the ReLU that produced the layer's input."""
    _check_bf16(x, w)
    x, lda = as_pixel_rows(x)
    M, K = x.shape
    N = w.shape[0]
    ldb = pixel_pitch(w)  # a row slice of a wider matrix (the transposed shadow's columns) is read in place
    if ldb is None:
        w = w.contiguous()
        ldb = K
    if out is None:  # rows padded to a multiple of 8 columns keep the 16-byte store path (N = 132, 4, ...)
        out = torch.empty(M, (N + 7) // 8 * 8, dtype=torch.bfloat16, device=x.device)[:, :N]
    ldc = pixel_pitch(out)
    if ldc is None:
        raise _lib.DfineHipError("linear_forward: output must be a pixel-row view")
    g = _geom(M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, gather=0, act=1 if relu else 0)
    if relu_mask is not None and bias is None and not relu:
        ldm = pixel_pitch(relu_mask)
        if ldm is not None and tuple(relu_mask.shape) == (M, N):
            lib = _lib.load()
            _lib.require_gpu(x, w, out, relu_mask)
            with _lib.timed(name + "_rows", (M * K + N * K + 2 * M * N) * 2, f"linear M{M} N{N} K{K} +relu mask", flops=2.0 * M * N * K):
                rc = lib.dfine_igemm_bf16_relumask(_lib.ptr(x), _lib.ptr(w), _lib.ptr(out), ctypes.byref(g), _lib.ptr(relu_mask), ldm,
                                                   _lib.stream_ptr())
            if rc == 0:
                return out
            if rc != -2:
                _lib.check(rc, "dfine_igemm_bf16_relumask")
        _launch(x, w, out, bias, None, g, name, f"linear M{M} N{N} K{K}")  # shapes off the 16-byte paths: GEMM, then the mask
        return torch.ops.aten.threshold_backward(out, relu_mask, 0)
    _launch(x, w, out, bias, None, g, name, f"linear M{M} N{N} K{K}")
    return out


def conv_out_hw(H, W, KH, KW, stride, pad, pad_br=None):
    pb = pad if pad_br is None else pad_br
    return (H + pad + pb - KH) // stride + 1, (W + pad + pb - KW) // stride + 1


def conv_forward(x, w_khwc, stride=1, pad=0, pad_br=None, out=None, stats=None, bias=None, relu=False):
    """x [B, H, W, CI] pixel rows, w_khwc [CO, KH, KW, CI] dense bf16 -> [B, OH, OW, CO] (``out`` may be a channel
    slice of a wider map). ``pad`` pads top/left, ``pad_br`` bottom/right (default: same). ``stats``: fp64-as-f32
    workspace [8*2*CO*2] that receives the BatchNorm sums (see include/dfine_hip.h)."""
    _check_bf16(x, w_khwc)
    x, lda = as_pixel_rows(x)
    B, H, W, CI = x.shape
    CO, KH, KW, _ = w_khwc.shape
    OH, OW = conv_out_hw(H, W, KH, KW, stride, pad, pad_br)
    if out is None:
        out = torch.empty(B, OH, OW, CO, dtype=torch.bfloat16, device=x.device)
    ldc = pixel_pitch(out)
    if ldc is None or tuple(out.shape) != (B, OH, OW, CO):
        raise _lib.DfineHipError("conv_forward: output must be a [B, OH, OW, CO] pixel-row view")
    plain = KH == 1 and KW == 1 and stride == 1 and pad == 0 and (pad_br or 0) == 0
    if plain:
        g = _geom(M=B * H * W, N=CO, K=CI, lda=lda, ldb=CI, ldc=ldc, gather=0, act=1 if relu else 0)
    else:
        taps = [(ky - pad, kx - pad, ky * KW + kx) for ky in range(KH) for kx in range(KW)]
        g = _geom(M=B * OH * OW, N=CO, K=KH * KW * CI, lda=lda, ldb=KH * KW * CI, ldc=ldc, gather=1, PH=OH, PW=OW, IH=H, IW=W,
                  CI=CI, isy=stride, isx=stride, OHf=OH, OWf=OW, osy=1, osx=1, ooy=0, oox=0, act=1 if relu else 0, taps=taps)
    _launch(x, w_khwc, out, bias, stats, g, "conv_fwd", f"conv {B}x{H}x{W} {CI}->{CO} k{KH}s{stride}")
    return out


def conv_input_grad(gy, wt_ckhwk, in_hw, stride=1, pad=0, out=None, accumulate=False, bnred=None):
    """gy [B, OH, OW, CO] pixel rows, wt [CI, KH, KW, CO] dense bf16 (the transposed weight) -> dx [B, H, W, CI].
    ``accumulate`` (stride 1, with ``out``): out += dx in the GEMM's epilogue. ``bnred`` (a BnRed, stride 1 only): the norm
    that produced this conv's input gets its backward sums from the launch's epilogue; ``out._bnred_done`` says whether it did."""
    _check_bf16(gy, wt_ckhwk)
    gy, lda = as_pixel_rows(gy)
    B, OH, OW, CO = gy.shape
    CI, KH, KW, _ = wt_ckhwk.shape
    H, W = in_hw
    if out is None:
        out = torch.empty(B, H, W, CI, dtype=torch.bfloat16, device=gy.device)
    ldc = pixel_pitch(out)
    if ldc is None or tuple(out.shape) != (B, H, W, CI):
        raise _lib.DfineHipError("conv_input_grad: output must be a [B, H, W, CI] pixel-row view")
    detail = f"dgrad {B}x{H}x{W} {CI}<-{CO} k{KH}s{stride}"
    acc = 0x400 if accumulate else 0
    if accumulate and stride != 1:
        raise _lib.DfineHipError("conv_input_grad: accumulate is for stride-1 convolutions")
    if bnred is not None and stride != 1:
        bnred = None
    if KH == 1 and KW == 1 and stride == 1 and pad == 0:
        g = _geom(M=B * H * W, N=CI, K=CO, lda=lda, ldb=CO, ldc=ldc, gather=0, act=acc)
        out._bnred_done = _launch(gy, wt_ckhwk, out, None, None, g, "conv_dgrad", detail, bnred)
        return out
    classes = []
    for ey in range(stride):
        for ex in range(stride):
            taps = [((ey + pad - ky) // stride, (ex + pad - kx) // stride, ky * KW + kx)
                    for ky in range(KH) for kx in range(KW)
                    if (ey + pad - ky) % stride == 0 and (ex + pad - kx) % stride == 0]
            ph, pw = (H - ey + stride - 1) // stride, (W - ex + stride - 1) // stride
            classes.append((ey, ex, ph, pw, taps))
    if any(not taps for *_, taps in classes):
        out.zero_()  # input pixels no output ever reads
    for ey, ex, ph, pw, taps in classes:
        if not taps or ph <= 0 or pw <= 0:
            continue
        g = _geom(M=B * ph * pw, N=CI, K=len(taps) * CO, lda=lda, ldb=KH * KW * CO, ldc=ldc, gather=1, PH=ph, PW=pw, IH=OH, IW=OW,
                  CI=CO, isy=1, isx=1, OHf=H, OWf=W, osy=stride, osx=stride, ooy=ey, oox=ex, taps=taps, act=acc)
        out._bnred_done = _launch(gy, wt_ckhwk, out, None, None, g, "conv_dgrad", detail, bnred)
    return out


def conv_transpose2x2_forward(x, wt_okkc, out=None, stats=None):
    """ConvTranspose2d(kernel 2, stride 2, no bias): x [B, H, W, CI], wt [CO, 2, 2, CI] (the TRANSPOSED copy of the
    module's [CI, 2, 2, CO] channels_last weight) -> [B, 2H, 2W, CO]."""
    _check_bf16(x, wt_okkc)
    x, lda = as_pixel_rows(x)
    B, H, W, CI = x.shape
    CO = wt_okkc.shape[0]
    if out is None:
        out = torch.empty(B, 2 * H, 2 * W, CO, dtype=torch.bfloat16, device=x.device)
    ldc = pixel_pitch(out)
    for ky in range(2):
        for kx in range(2):
            g = _geom(M=B * H * W, N=CO, K=CI, lda=lda, ldb=4 * CI, ldc=ldc, gather=1, PH=H, PW=W, IH=H, IW=W, CI=CI, isy=1, isx=1,
                      OHf=2 * H, OWf=2 * W, osy=2, osx=2, ooy=ky, oox=kx, taps=[(0, 0, ky * 2 + kx)])
            _launch(x, wt_okkc, out, None, stats, g, "conv_fwd", f"convT {B}x{H}x{W} {CI}->{CO}")
    return out


def conv_transpose2x2_input_grad(gy, w_ikkc, out=None):
    """gy [B, 2H, 2W, CO], w [CI, 2, 2, CO] (the module's channels_last weight) -> dx [B, H, W, CI]."""
    _check_bf16(gy, w_ikkc)
    gy, lda = as_pixel_rows(gy)
    B, H2, W2, CO = gy.shape
    CI = w_ikkc.shape[0]
    H, W = H2 // 2, W2 // 2
    if out is None:
        out = torch.empty(B, H, W, CI, dtype=torch.bfloat16, device=gy.device)
    ldc = pixel_pitch(out)
    taps = [(ky, kx, ky * 2 + kx) for ky in range(2) for kx in range(2)]
    g = _geom(M=B * H * W, N=CI, K=4 * CO, lda=lda, ldb=4 * CO, ldc=ldc, gather=1, PH=H, PW=W, IH=H2, IW=W2, CI=CO, isy=2, isx=2,
              OHf=H, OWf=W, osy=1, osx=1, ooy=0, oox=0, taps=taps)
    _launch(gy, w_ikkc, out, None, None, g, "conv_dgrad", f"convT dgrad {B}x{H}x{W} {CI}<-{CO}")
    return out
