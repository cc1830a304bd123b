"""Model-size tables for the MI355X build (data restated from the reference's
src/d_fine/configs.py:1-213 and the HGNetv2 stage table, src/d_fine/arch/hgnetv2.py:344-410).
"""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class StageSpec:
    c_in: int
    c_mid: int
    c_out: int
    n_blocks: int
    downsample: bool
    light: bool
    kernel: int
    n_layers: int


@dataclass(frozen=True)
class BackboneSpec:
    stem: Tuple[int, int, int]  # (in, mid, out)
    stages: Tuple[StageSpec, ...]


def _stages(rows):
    return tuple(StageSpec(*r) for r in rows)


BACKBONES = {
    "B0": BackboneSpec((3, 16, 16), _stages([(16, 16, 64, 1, False, False, 3, 3), (64, 32, 256, 1, True, False, 3, 3),
                                              (256, 64, 512, 2, True, True, 5, 3), (512, 128, 1024, 1, True, True, 5, 3)])),
    "B2": BackboneSpec((3, 24, 32), _stages([(32, 32, 96, 1, False, False, 3, 4), (96, 64, 384, 1, True, False, 3, 4),
                                              (384, 128, 768, 3, True, True, 5, 4), (768, 256, 1536, 1, True, True, 5, 4)])),
    "B4": BackboneSpec((3, 32, 48), _stages([(48, 48, 128, 1, False, False, 3, 6), (128, 96, 512, 1, True, False, 3, 6),
                                              (512, 192, 1024, 3, True, True, 5, 6), (1024, 384, 2048, 1, True, True, 5, 6)])),
    "B5": BackboneSpec((3, 32, 64), _stages([(64, 64, 128, 1, False, False, 3, 6), (128, 128, 512, 2, True, False, 3, 6),
                                              (512, 256, 1024, 5, True, True, 5, 6), (1024, 512, 2048, 2, True, True, 5, 6)])),
}


@dataclass(frozen=True)
class ModelSpec:
    name: str
    backbone: str
    return_idx: Tuple[int, ...]
    use_lab: bool
    frozen_backbone_norm: bool  # freeze_norm=True & freeze_at=0: FrozenBN everywhere + stem params frozen (l, x)
    in_channels: Tuple[int, ...]
    strides: Tuple[int, ...]
    enc_hidden: int
    enc_ffn: int
    expansion: float
    depth_mult: float
    hidden: int
    dec_ffn: int
    num_layers: int
    num_points: Tuple[int, ...]
    reg_scale: int  # stored as an int64 Parameter, like the reference
    # shared defaults (base_cfg + constructor defaults)
    num_heads: int = 8
    num_queries: int = 300
    num_denoising: int = 100
    label_noise_ratio: float = 0.5
    reg_max: int = 32
    mask_dim: int = 256
    anchor_eps: float = 1e-2
    anchor_grid: float = 0.05

    @property
    def csp_c4(self):  # round(expansion * hidden // 2), left to right (hybrid_encoder.py:384-385)
        return round(self.expansion * self.enc_hidden // 2)

    @property
    def csp_blocks(self):
        return round(3 * self.depth_mult)

    @property
    def num_levels(self):
        return len(self.in_channels)


MODELS = {
    "n": ModelSpec("n", "B0", (2, 3), True, False, (512, 1024), (16, 32), 128, 512, 0.34, 0.5, 128, 512, 3, (6, 6), 4),
    "s": ModelSpec("s", "B0", (1, 2, 3), True, False, (256, 512, 1024), (8, 16, 32), 256, 1024, 0.5, 0.34, 256, 1024, 3, (3, 6, 3), 4),
    "m": ModelSpec("m", "B2", (1, 2, 3), True, False, (384, 768, 1536), (8, 16, 32), 256, 1024, 1.0, 0.67, 256, 1024, 4, (3, 6, 3), 4),
    "l": ModelSpec("l", "B4", (1, 2, 3), False, True, (512, 1024, 2048), (8, 16, 32), 256, 1024, 1.0, 1.0, 256, 1024, 6, (3, 6, 3), 4),
    "x": ModelSpec("x", "B5", (1, 2, 3), False, True, (512, 1024, 2048), (8, 16, 32), 384, 2048, 1.0, 1.0, 256, 1024, 6, (3, 6, 3), 8),
}


@dataclass(frozen=True)
class CriterionSpec:
    """configs.py:25-51. The matcher's cost_mask / cost_mask_dice entries exist in the reference's
    config but are never read (matcher.py:40-42), so they are not carried."""
    weight_dict: dict = field(default_factory=lambda: dict(
        loss_vfl=1, loss_bbox=5, loss_giou=2, loss_fgl=0.15, loss_ddf=1.5, loss_mask_bce=10, loss_mask_dice=10))
    losses: Tuple[str, ...] = ("vfl", "boxes", "local")
    alpha: float = 0.75
    gamma: float = 2.0
    reg_max: int = 32
    cost_class: float = 2.0
    cost_bbox: float = 5.0
    cost_giou: float = 2.0
    match_alpha: float = 0.25
    match_gamma: float = 2.0
    ddf_temperature: float = 5.0
