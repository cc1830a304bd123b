"""Per-size hyper-parameters of D-FINE, restated from the reference as plain data
(reference src/d_fine/configs.py:1-213 and the HGNetv2 table src/d_fine/arch/hgnetv2.py:344-410).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""

# HGNetv2 variants: stem (in, mid, out) and per stage
# (in, mid, out, n_blocks, downsample, light_block, kernel, n_layers)   hgnetv2.py:344-398
HGNET = {
    "B0": dict(stem=(3, 16, 16), stages=[(16, 16, 64, 1, False, False, 3, 3), (64, 32, 256, 1, True, False, 3, 3),
                                          (256, 64, 512, 2, True, True, 5, 3), (512, 128, 1024, 1, True, True, 5, 3)]),
    "B2": dict(stem=(3, 24, 32), stages=[(32, 32, 96, 1, False, False, 3, 4), (96, 64, 384, 1, True, False, 3, 4),
                                          (384, 128, 768, 3, True, True, 5, 4), (768, 256, 1536, 1, True, True, 5, 4)]),
    "B4": dict(stem=(3, 32, 48), stages=[(48, 48, 128, 1, False, False, 3, 6), (128, 96, 512, 1, True, False, 3, 6),
                                          (512, 192, 1024, 3, True, True, 5, 6), (1024, 384, 2048, 1, True, True, 5, 6)]),
    "B5": dict(stem=(3, 32, 64), stages=[(64, 64, 128, 1, False, False, 3, 6), (128, 128, 512, 2, True, False, 3, 6),
                                          (512, 256, 1024, 5, True, True, 5, 6), (1024, 512, 2048, 2, True, True, 5, 6)]),
}

# configs.py:54-200 merged over base_cfg (1-52). `frozen` = freeze_at>=0 & freeze_norm (l, x).
SIZES = {
    "n": dict(backbone="B0", return_idx=(2, 3), use_lab=True, frozen=False, in_channels=(512, 1024), strides=(16, 32),
              enc_hidden=128, enc_ffn=512, expansion=0.34, depth_mult=0.5, hidden=128, dec_ffn=512, layers=3,
              points=(6, 6), reg_scale=4),
    "s": dict(backbone="B0", return_idx=(1, 2, 3), use_lab=True, frozen=False, in_channels=(256, 512, 1024),
              strides=(8, 16, 32), enc_hidden=256, enc_ffn=1024, expansion=0.5, depth_mult=0.34, hidden=256,
              dec_ffn=1024, layers=3, points=(3, 6, 3), reg_scale=4),
    "m": dict(backbone="B2", return_idx=(1, 2, 3), use_lab=True, frozen=False, in_channels=(384, 768, 1536),
              strides=(8, 16, 32), enc_hidden=256, enc_ffn=1024, expansion=1.0, depth_mult=0.67, hidden=256,
              dec_ffn=1024, layers=4, points=(3, 6, 3), reg_scale=4),
    "l": dict(backbone="B4", return_idx=(1, 2, 3), use_lab=False, frozen=True, in_channels=(512, 1024, 2048),
              strides=(8, 16, 32), enc_hidden=256, enc_ffn=1024, expansion=1.0, depth_mult=1.0, hidden=256,
              dec_ffn=1024, layers=6, points=(3, 6, 3), reg_scale=4),
    "x": dict(backbone="B5", return_idx=(1, 2, 3), use_lab=False, frozen=True, in_channels=(512, 1024, 2048),
              strides=(8, 16, 32), enc_hidden=384, enc_ffn=2048, expansion=1.0, depth_mult=1.0, hidden=256,
              dec_ffn=1024, layers=6, points=(3, 6, 3), reg_scale=8),
}

# base_cfg (configs.py:1-52) and constructor defaults that matter on the path
NUM_HEADS = 8
NUM_QUERIES = 300
NUM_DENOISING = 100
REG_MAX = 32
MASK_DIM = 256
ANCHOR_EPS = 1e-2  # dfine_decoder.py:561
LOSS_WEIGHTS = dict(loss_vfl=1, loss_bbox=5, loss_giou=2, loss_fgl=0.15, loss_ddf=1.5, loss_mask_bce=10, loss_mask_dice=10)
MATCH_WEIGHTS = dict(cost_class=2, cost_bbox=5, cost_giou=2)  # cost_mask* exist in the config but are never read (matcher.py:40-42)
MATCH_ALPHA, MATCH_GAMMA = 0.25, 2.0
VFL_ALPHA, VFL_GAMMA = 0.75, 2.0


def size_cfg(name):
    c = dict(SIZES[name])
    c["name"] = name
    c["csp_c4"] = round(c["expansion"] * c["enc_hidden"] // 2)  # hybrid_encoder.py:384-385
    c["csp_blocks"] = round(3 * c["depth_mult"])
    c["head_dim"] = c["hidden"] // NUM_HEADS
    return c
