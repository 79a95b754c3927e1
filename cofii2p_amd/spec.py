"""Architecture tables for the CoFiI2P hot path and the state_dict layout they imply.

The reference keeps its architecture as literals spread over model/network.py:21-43,
model/kpconv/kp_backbone.py:8-77 and model/imagenet.py:119-250.  Here it is ONE table that
drives (i) the parameter tree of `cofii2p_amd.network.CoFiI2P` (so released ``.t7`` files load
with ``strict=True``, eval_all.py:49), (ii) the weight packer and (iii) the execution plan.
`tests/test_spec.py` pins the generated key/shape list against the list dumped from the
reference (tests/golden/state_dict_spec.json).
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional, Tuple

KP_K = 15  # kernel points per KPConv (network.py:24 kernel_size=15)
KP_INIT_RADIUS = 4.25 * 0.1  # network.py:24
KP_INIT_SIGMA = 2 * 0.1
GN_GROUPS = 32
NUM_NEIGHBORS = 128  # preprocess_data.py:38 radius_num
D_MODEL = 128
N_HEAD = 4
N_LAYERS = 8
LAYER_KINDS = ("self", "cross") * 4  # network.py:35


@dataclass(frozen=True)
class KPBlock:
    """One block of the KPConv-FPN encoder (kp_backbone.py:11-73)."""

    name: str
    kind: str  # "conv" (ConvBlock) | "res" (ResidualBlock)
    cin: int
    cout: int
    stage: int  # query stage index (0 = 20480 points)
    strided: bool  # support = stage-1 points, indices = subsampling[stage-1]
    scale: int  # radius / sigma multiplier

    @property
    def mid(self) -> int:
        return self.cout // 4 if self.kind == "res" else self.cout

    @property
    def sigma(self) -> float:
        return KP_INIT_SIGMA * self.scale

    @property
    def radius(self) -> float:
        return KP_INIT_RADIUS * self.scale

    @property
    def has_shortcut_unary(self) -> bool:
        return self.kind == "res" and self.cin != self.cout


ENCODER: Tuple[KPBlock, ...] = (
    KPBlock("encoder1_1", "conv", 4, 64, 0, False, 1),
    KPBlock("encoder1_2", "res", 64, 128, 0, False, 1),
    KPBlock("encoder2_1", "res", 128, 128, 1, True, 1),
    KPBlock("encoder2_2", "res", 128, 256, 1, False, 2),
    KPBlock("encoder2_3", "res", 256, 256, 1, False, 2),
    KPBlock("encoder3_1", "res", 256, 256, 2, True, 2),
    KPBlock("encoder3_2", "res", 256, 512, 2, False, 4),
    KPBlock("encoder3_3", "res", 512, 512, 2, False, 4),
    KPBlock("encoder4_1", "res", 512, 512, 3, True, 4),
    KPBlock("encoder4_2", "res", 512, 1024, 3, False, 8),
    KPBlock("encoder4_3", "res", 1024, 1024, 3, False, 8),
    KPBlock("encoder5_1", "res", 1024, 1024, 4, True, 8),
    KPBlock("encoder5_2", "res", 1024, 2048, 4, False, 16),
    KPBlock("encoder5_3", "res", 2048, 2048, 4, False, 16),
)
# decoders (kp_backbone.py:75-77): name, in, out, has GroupNorm+LeakyReLU
DECODERS = (("decoder4", 3072, 1024, True), ("decoder3", 1536, 512, True), ("decoder2", 768, 64, False))

RESNET_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # planes, blocks, stride


def _f32(shape):
    return (tuple(shape), "float32")


def state_dict_spec(norm: str = "gn") -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """name -> (shape, dtype) in the reference's registration order.  norm: the point encoder's normalisation, get_norm() of
    model/kpconv/modules.py:51-60: 'gn' GroupNorm(32, C) (the shipped configuration), 'bn' BatchNorm1d(C), 'ln' LayerNorm(C)."""
    if norm not in ("gn", "bn", "ln"):
        raise ValueError("only support batch normalization, layer normalization and group normalization now!")   # modules.py:60
    s: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()

    # --- image encoder: ResNet-34 with affine-less InstanceNorm (imagenet.py:119-217) ---
    bb = "img_encoder.backbone."
    s[bb + "conv1.weight"] = _f32((64, 3, 7, 7))
    inpl = 64
    for li, (planes, blocks, stride) in enumerate(RESNET_LAYERS, start=1):
        for b in range(blocks):
            p = "%slayer%d.%d." % (bb, li, b)
            s[p + "conv1.weight"] = _f32((planes, inpl if b == 0 else planes, 3, 3))
            s[p + "conv2.weight"] = _f32((planes, planes, 3, 3))
            if b == 0 and (stride != 1 or inpl != planes):
                s[p + "downsample.0.weight"] = _f32((planes, inpl, 1, 1))
        inpl = planes
    s[bb + "fc.weight"] = _f32((1000, 512))  # never used by forward; kept for strict loading
    s[bb + "fc.bias"] = _f32((1000,))

    # --- point encoder: KPConv-FPN (kp_backbone.py) ---
    def pc_norm(prefix, c):   # prefix ends in "norm." / "norm_conv."
        if norm == "gn":      # modules.py:32-49: wrapper module with the nn.GroupNorm as `.norm`
            s[prefix + "norm.weight"] = _f32((c,))
            s[prefix + "norm.bias"] = _f32((c,))
        else:
            s[prefix + "weight"] = _f32((c,))
            s[prefix + "bias"] = _f32((c,))
            if norm == "bn":
                s[prefix + "running_mean"] = _f32((c,))
                s[prefix + "running_var"] = _f32((c,))
                s[prefix + "num_batches_tracked"] = ((), "int64")

    def unary(prefix, cin, cout, has_norm=True):
        s[prefix + "mlp.weight"] = _f32((cout, cin))
        s[prefix + "mlp.bias"] = _f32((cout,))
        if has_norm:
            pc_norm(prefix + "norm.", cout)

    for blk in ENCODER:
        p = "pc_encoder.%s." % blk.name
        if blk.kind == "conv":
            s[p + "KPConv.weights"] = _f32((KP_K, blk.cin, blk.cout))
            s[p + "KPConv.bias"] = _f32((blk.cout,))
            s[p + "KPConv.kernel_points"] = _f32((KP_K, 3))
            pc_norm(p + "norm.", blk.cout)
        else:
            mid = blk.mid
            if blk.cin != mid:
                unary(p + "unary1.", blk.cin, mid)
            s[p + "KPConv.weights"] = _f32((KP_K, mid, mid))
            s[p + "KPConv.bias"] = _f32((mid,))
            s[p + "KPConv.kernel_points"] = _f32((KP_K, 3))
            pc_norm(p + "norm_conv.", mid)
            unary(p + "unary2.", mid, blk.cout)
            if blk.has_shortcut_unary:
                unary(p + "unary_shortcut.", blk.cin, blk.cout)
    for name, cin, cout, has_norm in DECODERS:
        unary("pc_encoder.%s." % name, cin, cout, has_norm)

    # --- heads (network.py:29-43) ---
    s["pc_feature_layer.0.weight"] = _f32((1024, 2048))
    s["pc_feature_layer.1.weight"] = _f32((1024,))
    s["pc_feature_layer.1.bias"] = _f32((1024,))
    s["pc_feature_layer.3.weight"] = _f32((512, 1024))
    s["pc_feature_layer.4.weight"] = _f32((512,))
    s["pc_feature_layer.4.bias"] = _f32((512,))
    s["pc_feature_layer.6.weight"] = _f32((128, 512))
    for i in (0, 3, 6):  # unused by forward
        s["img_feature_layer.%d.weight" % i] = _f32((128, 128, 1, 1))
    for l in range(N_LAYERS):
        p = "transformer.layers.%d." % l
        for n in ("q_proj", "k_proj", "v_proj", "merge"):
            s[p + n + ".weight"] = _f32((D_MODEL, D_MODEL))
        s[p + "mlp.0.weight"] = _f32((2 * D_MODEL, 2 * D_MODEL))
        s[p + "mlp.2.weight"] = _f32((D_MODEL, 2 * D_MODEL))
        for n in ("norm1", "norm2"):
            s[p + n + ".weight"] = _f32((D_MODEL,))
            s[p + n + ".bias"] = _f32((D_MODEL,))
    for name, nd in (("fine_img_pos_encoding", 2), ("fine_pc_pos_encoding", 3)):  # unused by forward
        dims = (nd, 32, 64, 128, 256, 64)
        for j in range(5):
            s["%s.mlp.%d.weight" % (name, 2 * j)] = _f32((dims[j + 1], dims[j]))
            s["%s.mlp.%d.bias" % (name, 2 * j)] = _f32((dims[j + 1],))

    def bn(prefix, c):
        s[prefix + "weight"] = _f32((c,))
        s[prefix + "bias"] = _f32((c,))
        s[prefix + "running_mean"] = _f32((c,))
        s[prefix + "running_var"] = _f32((c,))
        s[prefix + "num_batches_tracked"] = ((), "int64")

    for name, cin, cout in (("img_upsample_1", 192, 128), ("img_upsample_2", 192, 64)):
        for j, ci in enumerate((cin, cout)):
            p = "%s.conv.%d." % (name, j)
            s[p + "conv1.weight"] = _f32((cout, ci, 3, 3))
            bn(p + "bn1.", cout)
            s[p + "conv2.weight"] = _f32((cout, cout, 3, 3))
            bn(p + "bn2.", cout)
            s[p + "conv_skip.0.weight"] = _f32((cout, ci, 3, 3))
            bn(p + "conv_skip.1.", cout)
    for i, (co, ci) in zip((0, 3, 6), ((128, 128), (64, 128), (1, 64))):
        s["pc_score_layer.%d.weight" % i] = _f32((co, ci, 1))
    for i, (co, ci) in zip((0, 3, 6), ((128, 128), (64, 128), (1, 64))):
        s["img_score_layer.%d.weight" % i] = _f32((co, ci, 1, 1))
    return s


BUFFER_LEAVES = ("kernel_points", "running_mean", "running_var", "num_batches_tracked")


def is_buffer(name: str) -> bool:
    return name.rsplit(".", 1)[-1] in BUFFER_LEAVES


def kpconv_radius_of(name: str) -> Optional[float]:
    for blk in ENCODER:
        if name.startswith("pc_encoder.%s." % blk.name):
            return blk.radius
    return None


def synth_state_dict(salt: int = 0, norm: str = "gn"):
    """Full synthetic state_dict (numpy arrays) from the name-keyed generator."""
    import numpy as np

    from .weights import kernel_point_table, synth_tensor

    out = OrderedDict()
    for name, (shape, dtype) in state_dict_spec(norm).items():
        if name.endswith("kernel_points"):
            out[name] = kernel_point_table(KP_K, kpconv_radius_of(name))
        else:
            out[name] = synth_tensor(name, shape, dtype, salt)
        assert out[name].shape == tuple(shape), name
    return out


def stage_sizes(num_points: int, num_stages: int = 5) -> List[int]:
    """preprocess_data.py:55-69 — each stage keeps N//2 of the previous one."""
    out = [int(num_points)]
    for _ in range(num_stages - 1):
        out.append(out[-1] // 2)
    return out
