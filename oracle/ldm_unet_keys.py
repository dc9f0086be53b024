"""TEST INFRASTRUCTURE -- the parameter names and shapes of the UNet inside a single-file Stable-Diffusion checkpoint
(`model.diffusion_model.*`, the LDM / CompVis layout that `StableDiffusionPipeline.from_single_file` reads for the
reference, /root/reference/model_util.py:75-101,179-197).

This is a restatement of the CONSTRUCTOR of the public LDM UNet -- `UNetModel.__init__` of CompVis/stable-diffusion
`ldm/modules/diffusionmodules/openaimodel.py` (SD1.x / SD2.x) and Stability-AI/generative-models
`sgm/modules/diffusionmodules/openaimodel.py` (SDXL), with `ResBlock`, `Downsample`, `Upsample`, `SpatialTransformer`,
`BasicTransformerBlock`, `CrossAttention`, `FeedForward(glu=True)` from the same trees -- walked in module-registration
order, so the list is what `state_dict()` of that model holds.  It shares no code with `leco_amd/ckpt_convert.py`, whose
key map is derived from the DIFFUSERS block structure: the test (tests/test_reference_crosscheck.py /
tests/test_host.py) checks that converter against this list, not against its own inverse.

Pinned by public numbers: 686 / 686 / 1680 tensors and 859 520 964 / 865 910 724 / 2 567 463 684 parameters for
SD1.5 / SD2.1 / SDXL-base (the same totals the diffusers-layout models have).

`python oracle/ldm_unet_keys.py` rewrites tests/golden/ldm_unet_keys.json (names + shapes, no weights).
"""
import json
import os

# the `unet_config.params` of the public inference yamls (v1-inference.yaml, v2-inference-v.yaml, sd_xl_base.yaml)
CONFIGS = {
    "sd15": dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1), num_res_blocks=2,
                 channel_mult=(1, 2, 4, 4), num_heads=8, transformer_depth=1, context_dim=768, use_linear_in_transformer=False),
    "sd21": dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1), num_res_blocks=2,
                 channel_mult=(1, 2, 4, 4), num_head_channels=64, transformer_depth=1, context_dim=1024,
                 use_linear_in_transformer=True),
    "sdxl": dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2), num_res_blocks=2,
                 channel_mult=(1, 2, 4), num_head_channels=64, transformer_depth=(1, 2, 10), context_dim=2048,
                 use_linear_in_transformer=True, adm_in_channels=2816, num_classes="sequential"),
}
PUBLIC_COUNTS = {"sd15": (686, 859_520_964), "sd21": (686, 865_910_724), "sdxl": (1680, 2_567_463_684)}
PREFIX = "model.diffusion_model."


class _Keys:
    def __init__(self):
        self.items = []          # (name, shape) in registration order

    def add(self, name, *shape):
        self.items.append((name, tuple(int(s) for s in shape)))

    # --- leaf layers ---------------------------------------------------------------------------------------
    def conv(self, name, cin, cout, k):
        self.add(name + ".weight", cout, cin, k, k)
        self.add(name + ".bias", cout)

    def linear(self, name, cin, cout, bias=True):
        self.add(name + ".weight", cout, cin)
        if bias:
            self.add(name + ".bias", cout)

    def norm(self, name, c):     # GroupNorm32 / LayerNorm: affine
        self.add(name + ".weight", c)
        self.add(name + ".bias", c)

    # --- openaimodel.ResBlock: in_layers = [GroupNorm32, SiLU, conv3x3]; emb_layers = [SiLU, linear];
    #     out_layers = [GroupNorm32, SiLU, Dropout, zero_module(conv3x3)]; skip_connection = Identity | conv1x1
    def resblock(self, name, ch, emb, out):
        self.norm(name + ".in_layers.0", ch)
        self.conv(name + ".in_layers.2", ch, out, 3)
        self.linear(name + ".emb_layers.1", emb, out)
        self.norm(name + ".out_layers.0", out)
        self.conv(name + ".out_layers.3", out, out, 3)
        if out != ch:
            self.conv(name + ".skip_connection", ch, out, 1)

    # --- attention.CrossAttention: to_q / to_k / to_v without bias, to_out = [Linear, Dropout]
    def cross_attention(self, name, query_dim, context_dim, inner):
        self.linear(name + ".to_q", query_dim, inner, bias=False)
        self.linear(name + ".to_k", context_dim, inner, bias=False)
        self.linear(name + ".to_v", context_dim, inner, bias=False)
        self.linear(name + ".to_out.0", inner, query_dim)

    # --- attention.BasicTransformerBlock registers attn1, ff, attn2, norm1, norm2, norm3 (in this order);
    #     FeedForward(glu=True): net = [GEGLU(dim, 4 dim) with .proj = Linear(dim, 8 dim), Dropout, Linear(4 dim, dim)]
    def transformer_block(self, name, dim, context_dim):
        self.cross_attention(name + ".attn1", dim, dim, dim)
        self.linear(name + ".ff.net.0.proj", dim, 8 * dim)
        self.linear(name + ".ff.net.2", 4 * dim, dim)
        self.cross_attention(name + ".attn2", dim, context_dim, dim)
        for n in ("norm1", "norm2", "norm3"):
            self.norm(name + "." + n, dim)

    # --- attention.SpatialTransformer: norm, proj_in (conv1x1 | Linear), transformer_blocks, proj_out (zero_module)
    def spatial_transformer(self, name, ch, depth, context_dim, use_linear):
        self.norm(name + ".norm", ch)
        if use_linear:
            self.linear(name + ".proj_in", ch, ch)
        else:
            self.conv(name + ".proj_in", ch, ch, 1)
        for d in range(depth):
            self.transformer_block(f"{name}.transformer_blocks.{d}", ch, context_dim)
        if use_linear:
            self.linear(name + ".proj_out", ch, ch)
        else:
            self.conv(name + ".proj_out", ch, ch, 1)


def unet_keys(model_channels, channel_mult, num_res_blocks, attention_resolutions, transformer_depth, context_dim,
              use_linear_in_transformer, in_channels=4, out_channels=4, adm_in_channels=None, num_classes=None, **_heads):
    """`UNetModel.__init__`: returns [(name without prefix, shape)] in state_dict order.  (Head counts only reshape
    activations; inner_dim = n_heads * d_head = ch in every released model.)"""
    K = _Keys()
    nlev = len(channel_mult)
    depth = list(transformer_depth) if isinstance(transformer_depth, (tuple, list)) else [transformer_depth] * nlev
    nres = list(num_res_blocks) if isinstance(num_res_blocks, (tuple, list)) else [num_res_blocks] * nlev
    emb = 4 * model_channels
    K.linear("time_embed.0", model_channels, emb)
    K.linear("time_embed.2", emb, emb)
    if num_classes == "sequential":          # sgm: label_emb = Sequential(Sequential(linear, SiLU, linear))
        K.linear("label_emb.0.0", adm_in_channels, emb)
        K.linear("label_emb.0.2", emb, emb)
    K.conv("input_blocks.0.0", in_channels, model_channels, 3)
    chans = [model_channels]
    ch, ds, i = model_channels, 1, 1
    for level, mult in enumerate(channel_mult):
        for _ in range(nres[level]):
            K.resblock(f"input_blocks.{i}.0", ch, emb, mult * model_channels)
            ch = mult * model_channels
            if ds in attention_resolutions:
                K.spatial_transformer(f"input_blocks.{i}.1", ch, depth[level], context_dim, use_linear_in_transformer)
            chans.append(ch)
            i += 1
        if level != nlev - 1:
            K.conv(f"input_blocks.{i}.0.op", ch, ch, 3)            # Downsample(use_conv=True).op: 3x3 stride 2
            chans.append(ch)
            i += 1
            ds *= 2
    K.resblock("middle_block.0", ch, emb, ch)
    K.spatial_transformer("middle_block.1", ch, depth[-1], context_dim, use_linear_in_transformer)
    K.resblock("middle_block.2", ch, emb, ch)
    i = 0
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for j in range(nres[level] + 1):
            ich = chans.pop()
            K.resblock(f"output_blocks.{i}.0", ch + ich, emb, model_channels * mult)
            ch = model_channels * mult
            nxt = 1
            if ds in attention_resolutions:
                K.spatial_transformer(f"output_blocks.{i}.1", ch, depth[level], context_dim, use_linear_in_transformer)
                nxt = 2
            if level and j == nres[level]:
                K.conv(f"output_blocks.{i}.{nxt}.conv", ch, ch, 3)  # Upsample(use_conv=True).conv
                ds //= 2
            i += 1
    K.norm("out.0", ch)
    K.conv("out.2", model_channels, out_channels, 3)
    return K.items


def ldm_unet_keys(arch: str):
    """{full checkpoint key: shape} of the released `arch` in ('sd15', 'sd21', 'sdxl')."""
    return {PREFIX + n: list(s) for n, s in unet_keys(**CONFIGS[arch])}


def numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


if __name__ == "__main__":
    out = {}
    for arch in CONFIGS:
        keys = ldm_unet_keys(arch)
        n, p = len(keys), sum(numel(s) for s in keys.values())
        assert (n, p) == PUBLIC_COUNTS[arch], (arch, n, p, PUBLIC_COUNTS[arch])
        out[arch] = keys
        print(arch, n, "tensors", p, "parameters")
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ldm_unet_keys.json")
    with open(dst, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", os.path.normpath(dst))
