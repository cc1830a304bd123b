"""Building blocks shared by backbone / encoder / decoder. Parameter and buffer names (and
therefore ``state_dict`` keys) follow the reference so checkpoints are interchangeable
(SURVEY.md A.10); tensors are channel-last (NHWC memory) end to end.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class FrozenBatchNorm2d(nn.Module):
    """Affine + statistics as fixed buffers (reference src/d_fine/arch/common.py:29-67; eps inside
    the rsqrt). Used for every backbone norm of the l/x sizes."""

    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.eps = eps

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "num_batches_tracked", None)  # plain-BN checkpoints load too
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.view(1, -1, 1, 1).to(x.dtype) + shift.view(1, -1, 1, 1).to(x.dtype)


class LearnableAffine(nn.Module):
    """Scalar scale*x + bias after the activation (hgnetv2.py:25-32)."""

    def __init__(self):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor([1.0]))
        self.bias = nn.Parameter(torch.tensor([0.0]))

    def forward(self, x):
        return torch.addcmul(self.bias.to(x.dtype), x, self.scale.to(x.dtype))


_ACTS = {None: None, "relu": F.relu, "silu": F.silu}


class ConvNorm(nn.Module):
    """conv (no bias) -> batch norm -> activation -> optional learnable affine.

    One class for the reference's ConvBNAct (norm attribute ``bn``, hgnetv2.py:35-80) and
    ConvNormLayer / ConvNormLayer_fuse (attribute ``norm``, hybrid_encoder.py:21-93)."""

    def __init__(self, c_in, c_out, k, stride=1, groups=1, act=None, lab=False, norm_name="norm",
                 frozen_norm=False, padding=None):
        super().__init__()
        self.conv = nn.Conv2d(c_in, c_out, k, stride, (k - 1) // 2 if padding is None else padding,
                              groups=groups, bias=False)
        self.norm_name = norm_name
        setattr(self, norm_name, FrozenBatchNorm2d(c_out) if frozen_norm else nn.BatchNorm2d(c_out))
        self.act_name = act
        self.pad_br = None  # bottom/right padding when it differs from conv.padding (the stem's 2x2 convs)
        if lab and act is not None:
            self.lab = LearnableAffine()
        self._pending_batches = 0
        self.register_state_dict_pre_hook(ConvNorm._flush_batch_counter)

    deploy_fusable = False  # set by the encoder on the layers the reference builds as ConvNormLayer_fuse

    def convert_to_deploy(self):
        """Inference re-parameterisation (reference ConvNormLayer_fuse.convert_to_deploy, hybrid_encoder.py:47-79): fold
        the norm's running statistics and affine into the conv, which gains a bias; ``conv`` and the norm are removed."""
        if not self.deploy_fusable or hasattr(self, "conv_bn_fused"):
            return
        kernel, bias = fuse_conv_norm(self.conv, getattr(self, self.norm_name))
        c = self.conv
        fused = nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding, groups=c.groups, bias=True)
        fused = fused.to(kernel.device).to(memory_format=torch.channels_last)
        fused.weight.data = kernel.contiguous(memory_format=torch.channels_last)
        fused.bias.data = bias
        self.conv_bn_fused = fused
        delattr(self, "conv")
        delattr(self, self.norm_name)

    def forward(self, x, addend=None, act=None, into=None, post=None):
        """``addend`` (same shape as the output) is added after the norm, before the activation;
        ``act`` overrides the constructor's activation name (used by RepVgg: silu(a + b)); ``into`` (an ops.ConcatBuffer):
        the output becomes the buffer's next channel slice -- written there directly by the fused norm kernel; ``post``
        (same shape as the output) is added last, after activation and LAB (the HG block's residual), by the same kernel."""
        if into is not None:
            norm = getattr(self, self.norm_name, None)
            if (x.is_cuda and isinstance(norm, (nn.BatchNorm2d, nn.SyncBatchNorm, FrozenBatchNorm2d))
                    and ops.batch_norm_act_supported(x.to(torch.bfloat16) if torch.is_autocast_enabled() else x, self.out_channels)):
                # the fused norm kernel writes the slice
                return self._forward(x, addend, act, (into, into.take(self.out_channels)), post)
            return ops.concat_copy_in(self._forward(x, addend, act, None, post), into)  # deployed conv etc.: compute, then copy in
        return self._forward(x, addend, act, None, post)

    @property
    def out_channels(self):
        return (self.conv_bn_fused if hasattr(self, "conv_bn_fused") else self.conv).out_channels

    def _forward(self, x, addend, act, slot, post=None):
        """``slot`` = (buffer, first channel): only the fused-norm path can honour it; any other path returns None then."""
        if hasattr(self, "conv_bn_fused"):  # deployed: conv + bias (+ activation)
            y = ops.conv_bias_act(x, self.conv_bn_fused, act if act is not None else self.act_name, addend)
            y = self.lab(y) if hasattr(self, "lab") else y
            return y if post is None else y + post
        y, ws, have_stats = self._conv(x)
        return self._norm_tail(y, ws, have_stats, addend, act, slot, post)

    def _pairable(self, x):
        """This layer's conv can be one half of ops.dense_conv_pair: plain dense stride-1 conv on the bf16 GPU path."""
        if hasattr(self, "conv_bn_fused"):
            return False
        conv = self.conv
        return (conv.groups == 1 and conv.stride == (1, 1) and conv.padding[0] == conv.padding[1] and isinstance(conv.padding[0], int)
                and conv.dilation == (1, 1) and self.pad_br is None and x.is_cuda
                and (torch.is_autocast_enabled() or x.dtype == torch.bfloat16))

    def _stats_workspace(self, x):
        """(ws, have_stats): the norm's workspace when the conv's epilogue is to leave the batch statistics in it."""
        norm = getattr(self, self.norm_name)
        bf16_mode = x.is_cuda and (torch.is_autocast_enabled() or x.dtype == torch.bfloat16)
        if bf16_mode and self.training and isinstance(norm, (nn.BatchNorm2d, nn.SyncBatchNorm)) and norm.training:
            return ops.bn_workspace(self.conv.out_channels, x.device), True
        return None, False

    def _conv(self, x, weight=None):
        """The convolution alone -> (output, norm workspace, whether the workspace already holds the batch statistics).
        ``weight``: a stand-in for ``conv.weight`` with the same geometry (the stem's weight zero-padded to 4 input channels)."""
        conv = self.conv
        ws, have_stats = None, False
        if conv.groups == conv.in_channels == conv.out_channels and conv.groups > 1:
            if ops.depthwise_stats_supported(conv.stride[0]) and x.dtype == torch.bfloat16:
                ws, have_stats = self._stats_workspace(x)  # the strip kernel's epilogue leaves the statistics, like a dense conv's
            x = ops.depthwise_conv2d(x, conv.weight, conv.stride[0],  # HIP: MIOpen has no fast path here
                                     stats=ws[: 32 * conv.out_channels] if have_stats else None)
        elif (conv.groups == 1 and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]
              and isinstance(conv.padding[0], int) and conv.dilation == (1, 1)):
            # HIP implicit-GEMM forward / input gradient, HIP wgrad straight into the fp32 grads; in bf16 training mode
            # the conv's epilogue leaves the BatchNorm statistics in the norm's workspace (no statistics pass)
            ws, have_stats = self._stats_workspace(x)
            x = ops.dense_conv(x, conv.weight if weight is None else weight, conv.stride[0], conv.padding[0], self.pad_br,
                               stats=ws[: 32 * conv.out_channels] if have_stats else None)
        else:
            if self.pad_br is not None:
                x = F.pad(x, (0, self.pad_br, 0, self.pad_br))
            x = conv(x)
        return x, ws, have_stats

    def _norm_tail(self, x, ws, have_stats, addend=None, act=None, slot=None, post=None):
        """norm (+ addend) + activation (+ LAB) of a conv output."""
        conv = self.conv
        norm = getattr(self, self.norm_name)
        frozen = isinstance(norm, FrozenBatchNorm2d)
        # batch statistics as nn.BatchNorm2d decides: a norm put in .eval() inside a training model (frozen-statistics
        # fine-tuning) normalises with its running statistics and leaves them alone
        batch_stats = self.training and not frozen and norm.training
        act_name = act if act is not None else self.act_name
        lab = getattr(self, "lab", None)
        fused = (isinstance(norm, (nn.BatchNorm2d, nn.SyncBatchNorm)) or frozen) and ops.batch_norm_act_supported(x, conv.out_channels)
        if fused:  # HIP: norm + addend + activation + LAB in 1-2 kernels (2 more in backward); frozen/eval: 1 + 1
            if batch_stats:
                self._pending_batches += 1
            sync = None
            if batch_stats and isinstance(norm, nn.SyncBatchNorm):  # reference train.py:159-160: per-GPU batch < 4 under DDP
                sync = sync_group_of(norm)
            return ops.batch_norm_act(
                x, norm.weight, norm.bias, norm.running_mean, norm.running_var, act_name,
                lab.scale if lab is not None else None, lab.bias if lab is not None else None, addend,
                training=batch_stats, momentum=getattr(norm, "momentum", 0.1) or 0.1, eps=norm.eps, ws=ws, have_stats=have_stats,
                slot=slot, post=post, sync=sync)
        if slot is not None:
            raise ops._lib.DfineHipError("ConvNorm: a concat slot needs the fused norm kernel")
        x = norm(x)
        if addend is not None:
            x = x + addend
        if act_name is not None:
            x = _ACTS[act_name](x)
        if lab is not None:
            x = lab(x)
        return x if post is None else x + post

    def _flush_batch_counter(self, *args):
        # nn.BatchNorm2d counts forward calls in ``num_batches_tracked``; the fused path counts on the
        # host and writes the buffer only when a state_dict is taken (no per-layer launch per step)
        norm = getattr(self, self.norm_name, None)  # None once deployed (norm folded into the conv)
        if self._pending_batches and getattr(norm, "num_batches_tracked", None) is not None:
            norm.num_batches_tracked += self._pending_batches
        self._pending_batches = 0


def sync_group_of(norm):
    """(process group, ranks) over which an nn.SyncBatchNorm shares its batch statistics, or None when there is nothing to
    share (no process group initialised, or a single rank) -- nn.SyncBatchNorm.forward then is plain batch norm too."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = norm.process_group  # None = the default group
    ranks = dist.get_world_size(group)
    return (group, ranks) if ranks > 1 else None


def conv_pair(a, b, x):
    """The convolutions of two ConvNorm layers that read the SAME input (RepVgg's 3x3 + 1x1 branches, CSPLayer's two 1x1
    projections) -> ((ya, ws_a, stats_a), (yb, ws_b, stats_b)) for their ``_norm_tail``. On the bf16 GPU path the pair is one
    autograd node: its backward forms ONE input gradient, the second implicit GEMM adding into the first's output
    (ops.dense_conv_pair) -- otherwise autograd sums the two with an extra kernel over the map."""
    if a._pairable(x) and b._pairable(x) and torch.is_grad_enabled():
        (ws_a, hs_a), (ws_b, hs_b) = a._stats_workspace(x), b._stats_workspace(x)
        ya, yb = ops.dense_conv_pair(x, a.conv.weight, b.conv.weight, a.conv.padding[0], b.conv.padding[0],
                                     ws_a[: 32 * a.conv.out_channels] if hs_a else None, ws_b[: 32 * b.conv.out_channels] if hs_b else None)
        return (ya, ws_a, hs_a), (yb, ws_b, hs_b)
    return a._conv(x), b._conv(x)


def fuse_conv_norm(conv, norm):
    """(kernel, bias) of the conv equivalent to ``norm(conv(x))`` in eval mode (hybrid_encoder.py:68-79)."""
    std = (norm.running_var + norm.eps).sqrt()
    t = (norm.weight / std).reshape(-1, 1, 1, 1)
    return (conv.weight * t).detach(), (norm.bias - norm.running_mean * norm.weight / std).detach()


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / state_dict keys) with an optional fused residual and clamp:
    ``norm(x, res, clamp)`` = ``layer_norm(clamp(x + res))``. On the GPU this is one HIP launch per direction
    in the activations' dtype (bf16 under autocast; the stock op would run in fp32 between two casts)."""

    def forward(self, x, res=None, clamp=None):
        D = self.normalized_shape[0]
        if len(self.normalized_shape) == 1 and self.elementwise_affine and x.is_cuda:
            dt = torch.bfloat16 if torch.is_autocast_enabled() else x.dtype
            if ops.add_layer_norm_supported(x.to(dt) if x.dtype != dt else x, D):
                return ops.add_layer_norm(x.to(dt), res, self.weight, self.bias, self.eps, clamp)
        if res is not None:
            x = x + res
        if clamp:
            x = x.clamp(-clamp, clamp)
        return super().forward(x)


class Linear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose bf16 training path goes through
    ``ops.linear``: library GEMMs for y and dx, the HIP split-M kernel for dW / db."""

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class MLP(nn.Module):
    """Linear stack with ReLU in between (dfine_decoder.py:33-46); keys ``layers.{i}``."""

    def __init__(self, d_in, d_hidden, d_out, n_layers):
        super().__init__()
        dims = [d_in] + [d_hidden] * (n_layers - 1) + [d_out]
        self.layers = nn.ModuleList(Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            # bias + ReLU in the GEMM epilogue; a ReLU's backward in the epilogue of the NEXT layer's input-gradient GEMM
            x = ops.linear(x, layer.weight, layer.bias, relu=i < last, relu_in=i > 0, mask_by_consumer=i < last)
        return x


class PackedSelfAttention(nn.Module):
    """Self-attention with nn.MultiheadAttention's parameter layout (``in_proj_weight``
    [3D, D], ``in_proj_bias``, ``out_proj.{weight,bias}``) where queries and keys share one
    input and values another: q = k = x + pos, v = x (hybrid_encoder.py:274-277,
    dfine_decoder.py:236-239). The QK projection is a single GEMM."""

    def __init__(self, dim, n_head):
        super().__init__()
        self.dim, self.n_head = dim, n_head
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = Linear(dim, dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, qk_in, v_in, blocked=None):
        """qk_in, v_in: [B, L, D]; blocked: optional bool [L, L], True = may NOT attend."""
        B, L, D = qk_in.shape
        H, dh = self.n_head, D // self.n_head
        qk = ops.linear(qk_in, self.in_proj_weight, self.in_proj_bias, 0, 2 * D)
        v = ops.linear(v_in, self.in_proj_weight, self.in_proj_bias, 2 * D, 3 * D)
        if ops.self_attention_supported(qk, v, H):  # HIP: reads q / k / v where the projections left them, no head permutes
            return self.out_proj(ops.self_attention(qk, v, blocked, H))
        q, k = qk.view(B, L, 2, H, dh).permute(2, 0, 3, 1, 4)
        v = v.view(B, L, H, dh).transpose(1, 2)
        allowed = None if blocked is None else ~blocked
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=allowed)
        return self.out_proj(out.transpose(1, 2).reshape(B, L, D))


def bias_for_prior(p):
    return -math.log((1 - p) / p)
