"""Cross-checks against the reference's OWN files, imported read-only from /root/reference through
the stub `diffusers` (build container only; skipped where /root/reference does not exist, e.g. the
GPU box).  Covers what can be pinned exactly without diffusers."""
import contextlib
import importlib
import io
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "stub_diffusers"))
    sys.path.insert(0, REF)
    mods = {n: importlib.import_module(n) for n in ("lora", "prompt_util", "config_util", "train_util")}
    yield mods
    sys.path.remove(REF)


def test_lora_network_names_shapes_and_save_format(ref, tmp_path):
    from safetensors.torch import load_file
    from leco_amd import model_util
    from leco_amd.lora import LoRANetwork
    from leco_amd.unet import UNet2DConditionModel
    from oracle import unet_ref as R
    with torch.device("meta"):
        ru = R.UNet2DConditionModel(R.tiny_config())
        m = UNet2DConditionModel(model_util.tiny_config())
    torch.manual_seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        rnet = ref["lora"].LoRANetwork(ru, rank=4, multiplier=1.0, alpha=1.0)
    torch.manual_seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        net = LoRANetwork(m, rank=4, multiplier=1.0, alpha=1.0)
    rs, s = rnet.state_dict(), net.state_dict()
    assert list(rs.keys()) == list(s.keys())
    for k in rs:
        assert rs[k].shape == s[k].shape, k
        assert torch.equal(rs[k], s[k].float().cpu()), f"init stream differs at {k}"
    assert len(rnet.prepare_optimizer_params()[0]["params"]) == len(net.prepare_optimizer_params()[0]["params"])
    a, b = tmp_path / "ref.safetensors", tmp_path / "ours.safetensors"
    rnet.save_weights(str(a), dtype=torch.bfloat16)
    net.save_weights(str(b), dtype=torch.bfloat16)
    fa, fb = load_file(str(a)), load_file(str(b))
    assert list(fa.keys()) == list(fb.keys())
    for k in fa:
        assert fa[k].dtype == fb[k].dtype == torch.bfloat16 and torch.equal(fa[k], fb[k]), k


@pytest.mark.parametrize("arch,n_mods,n_params,rank", [("sd15", 192, 1695744, 4), ("sdxl", 722, 42557440, 16)])
def test_lora_census_full_size(ref, arch, n_mods, n_params, rank):
    from leco_amd import model_util
    from leco_amd.lora import LoRANetwork
    from leco_amd.unet import UNet2DConditionModel
    with torch.device("meta"):
        m = UNet2DConditionModel(model_util.SYNTHETIC[arch]())
        with contextlib.redirect_stdout(io.StringIO()):
            net = LoRANetwork(m, rank=rank)
    assert len(net.unet_loras) == n_mods and net.numel == n_params


def test_c3lier_census_sd15(ref):
    """SD1.5 c3lier: 278 modules / 8 406 528 parameters at rank 8."""
    from leco_amd import model_util
    from leco_amd.lora import DEFAULT_TARGET_REPLACE, UNET_TARGET_REPLACE_MODULE_CONV, LoRANetwork
    from leco_amd.unet import UNet2DConditionModel
    with torch.device("meta"):
        m = UNet2DConditionModel(model_util.SYNTHETIC["sd15"]())
        with contextlib.redirect_stdout(io.StringIO()):
            net = LoRANetwork(m, rank=8, target_replace_modules=list(DEFAULT_TARGET_REPLACE) + list(UNET_TARGET_REPLACE_MODULE_CONV))
    assert len(net.unet_loras) == 278 and net.numel == 8406528


def test_training_method_filter_quirk_is_reproduced(ref):
    """lora.py:169-187 filters on the OUTER module name, so selfattn / xattn select nothing (SURVEY F-2)."""
    from leco_amd import model_util
    from leco_amd.lora import LoRANetwork
    from leco_amd.unet import UNet2DConditionModel
    from oracle import unet_ref as R
    for method in ("full", "noxattn", "innoxattn", "selfattn", "xattn"):
        with torch.device("meta"), contextlib.redirect_stdout(io.StringIO()):
            want = len(ref["lora"].LoRANetwork(R.UNet2DConditionModel(R.tiny_config()), rank=4, train_method=method).unet_loras)
            got = len(LoRANetwork(UNet2DConditionModel(model_util.tiny_config()), rank=4, train_method=method).unet_loras)
        assert got == want, method


def test_example_yaml_schemas_parse_identically(ref):
    from leco_amd import config_util, prompt_util
    ex = os.path.join(REF, "examples")
    for name in ("prompts.yaml", "cat_ears_prompts.yaml", "unreal_prompts.yaml"):
        a = ref["prompt_util"].load_prompts_from_yaml(os.path.join(ex, name))
        b = prompt_util.load_prompts_from_yaml(os.path.join(ex, name))
        assert [x.dict() for x in a] == [x.model_dump() for x in b]
    for name in ("config.yaml", "cat_ears_config.yaml", "unreal_config.yaml"):
        a = ref["config_util"].load_config_from_yaml(os.path.join(ex, name))
        b = config_util.load_config_from_yaml(os.path.join(ex, name))
        assert a.dict() == b.model_dump()


def test_loss_and_step_primitives_equal_reference(ref):
    from leco_amd import prompt_util, train_util
    g = torch.Generator().manual_seed(0)
    t, p, n, u = [torch.randn(2, 4, 8, 8, generator=g) for _ in range(4)]
    for action in ("erase", "enhance"):
        rs = ref["prompt_util"].PromptSettings(target="x", action=action, guidance_scale=1.7)
        s = prompt_util.PromptSettings(target="x", action=action, guidance_scale=1.7)
        a = ref["prompt_util"].PromptEmbedsPair(torch.nn.MSELoss(), None, None, None, None, rs)
        b = prompt_util.PromptEmbedsPair(torch.nn.MSELoss(), None, None, None, None, s)
        kw = dict(target_latents=t, positive_latents=p, neutral_latents=n, unconditional_latents=u)
        assert torch.equal(a.loss(**kw), b.loss(**kw))
    e1, e2 = torch.randn(1, 77, 8, generator=g), torch.randn(1, 77, 8, generator=g)
    assert torch.equal(ref["train_util"].concat_embeddings(e1, e2, 3), train_util.concat_embeddings(e1, e2, 3))

    class Toy:
        def __call__(self, x, t, encoder_hidden_states=None):
            class O:
                sample = x * 0.5 + encoder_hidden_states.mean() + float(t) * 1e-3
            return O

    from leco_amd.scheduler import DDIMScheduler
    from oracle.ddim_ref import DDIMSchedulerRef
    sa, sb = DDIMSchedulerRef(), DDIMScheduler()
    sa.set_timesteps(10)
    sb.set_timesteps(10)
    lat = torch.randn(2, 4, 8, 8, generator=g)
    emb = torch.randn(4, 77, 8, generator=g)
    with contextlib.redirect_stderr(io.StringIO()):
        da = ref["train_util"].diffusion(Toy(), sa, lat, emb, start_timesteps=0, total_timesteps=4, guidance_scale=3)
    db = train_util.diffusion(Toy(), sb, lat, emb, start_timesteps=0, total_timesteps=4, guidance_scale=3)
    assert torch.allclose(da, db, rtol=1e-5, atol=1e-6)
    torch.manual_seed(5)
    ra = ref["train_util"].get_random_resolution_in_bucket(512)
    torch.manual_seed(5)
    assert ra == train_util.get_random_resolution_in_bucket(512)
    torch.manual_seed(6)
    la = ref["train_util"].get_initial_latents(sa, 2, 512, 512, 1)
    torch.manual_seed(6)
    assert torch.equal(la, train_util.get_initial_latents(sb, 2, 512, 512, 1))
    assert torch.equal(ref["train_util"].get_add_time_ids(1024, 1024), train_util.get_add_time_ids(1024, 1024))


def test_non_lora_forward_patch_on_the_hip_unet_fails_loudly():
    """A leaf whose `forward` was re-assigned by something that is NOT a LoRA module cannot be honoured (the launch plans
    never call leaf modules): refused loudly rather than silently ignored."""
    from leco_amd import model_util
    from leco_amd.unet import UNet2DConditionModel
    m = UNet2DConditionModel(model_util.tiny_config())
    leaf = m.down_blocks[0].attentions[0].proj_in
    leaf.forward = lambda x: x
    with pytest.raises(RuntimeError, match="not a LoRA module"):
        m(torch.zeros(2, 4, 16, 16), torch.tensor(1), encoder_hidden_states=torch.zeros(2, 77, 64))


def test_reference_lora_network_applied_to_this_unet_is_adopted(ref):
    """SURVEY 8b seam 2, the reference's LoRA injection (lora.py:97-106): the reference's OWN `lora.LoRANetwork`, built on
    this package's UNet, patches `leaf.forward`; the engine adopts it -- its parameters become slab views, the products
    run fused in the GEMMs -- so the reference's loop body works unmodified on it: `with network:` on / off,
    `loss.backward()` filling the foreign parameters' .grad, the foreign torch optimizer stepping them, `save_weights`.
    Checked against the reference network patched onto the fp32 oracle UNet (emulator backend, fp32 compute mode)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from leco_amd import hip, model_util
    from leco_amd.lora import ForeignLoRANetwork
    from leco_amd.unet import UNet2DConditionModel
    from oracle import unet_ref as R
    hip._use_library(build_emu.build())
    ru = R.init_synthetic_(R.UNet2DConditionModel(R.tiny_config()), seed=1234)
    ru.requires_grad_(False)
    m = UNet2DConditionModel(model_util.tiny_config())
    m.load_state_dict(ru.state_dict())
    m.requires_grad_(False)                      # fp32 model -> fp32 compute mode: tight comparison
    nets = []
    for unet in (ru, m):
        torch.manual_seed(42)
        with contextlib.redirect_stdout(io.StringIO()):
            net = ref["lora"].LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0)
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():
            for l in net.unet_loras:
                l.lora_up.weight.copy_(torch.randn(l.lora_up.weight.shape, generator=g) * 0.05)
        nets.append(net)
    rnet, fnet = nets
    opt_r = torch.optim.AdamW(rnet.prepare_optimizer_params(), lr=1e-3)
    opt_f = torch.optim.AdamW(fnet.prepare_optimizer_params(), lr=1e-3)
    g = torch.Generator().manual_seed(9)
    x, ctx = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 77, 64, generator=g)       # 8x8 latents: fp32 emulation
    tgt = torch.randn(2, 4, 8, 8, generator=g)

    def rel(a, b):      # (at 8x8 the deepest level is 1x1: its GroupNorm output, hence some gradients, are exactly 0 in both)
        return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()
    for it in range(2):          # second iteration: parameters changed by the FOREIGN optimizer behind the engine's back
        with rnet:
            yr = ru(x, torch.tensor(500), encoder_hidden_states=ctx).sample
        with fnet:
            yf = m(x, torch.tensor(500), encoder_hidden_states=ctx).sample
        assert isinstance(m.engine().network, ForeignLoRANetwork)
        assert rel(yf, yr) < 1e-4, (it, rel(yf, yr))
        ((yr - tgt) ** 2).mean().backward()
        ((yf - tgt) ** 2).mean().backward()
        for a, b in zip(rnet.unet_loras, fnet.unet_loras):
            assert rel(b.lora_up.weight.grad, a.lora_up.weight.grad) < 1e-3
        opt_r.step(); opt_f.step()
        opt_r.zero_grad(); opt_f.zero_grad()
    # outside `with network:` the multiplier is 0: the frozen model
    y0 = m(x, torch.tensor(500), encoder_hidden_states=ctx).sample
    assert rel(y0, ru(x, torch.tensor(500), encoder_hidden_states=ctx).sample) < 1e-4
    sa, sb = rnet.state_dict(), fnet.state_dict()
    assert list(sa) == list(sb) and all(rel(sb[k], sa[k]) < 1e-4 for k in sa if "lora_up" in k)


# ---- N2: the prompt-encoding front end against the reference's OWN loader and glue -------------------------------------
def _clip_folder(tmp_path, layers, with_projection_2=False):
    """A real `transformers` CLIP text stack on disk in the diffusers folder layout (reduced width, synthetic BPE vocabulary):
    tokenizer/, text_encoder/ (+ tokenizer_2/, text_encoder_2/ with a projection head for the XL glue)."""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host import _write_synthetic_clip
    folder = str(tmp_path / f"clip{layers}")
    te = _write_synthetic_clip(folder, hidden=64, layers=layers)
    te.save_pretrained(os.path.join(folder, "text_encoder"))
    te2 = None
    if with_projection_2:
        from transformers import CLIPTextModelWithProjection
        shutil.copytree(os.path.join(folder, "tokenizer"), os.path.join(folder, "tokenizer_2"))
        torch.manual_seed(1)
        te2 = CLIPTextModelWithProjection(te.config.__class__(**{**te.config.to_dict(), "hidden_size": 128, "intermediate_size": 256,
                                                                  "num_attention_heads": 2, "projection_dim": 96}))
        te2.save_pretrained(os.path.join(folder, "text_encoder_2"))
    return folder


PROMPTS = ["van gogh", "", "a b c " * 60, "monet water lilies"]      # incl. the empty prompt and one that is truncated at 77


@pytest.mark.parametrize("v2,clip_skip", [(False, None), (False, 2), (True, None), (True, 3)])
def test_prompt_front_end_equals_the_reference_loader_and_glue(ref, tmp_path, monkeypatch, v2, clip_skip):
    """N2 pinned to the reference: ITS `model_util.load_diffusers_model` (model_util.py:29-72; only the tokenizer's hub name is
    pointed at the local folder and the stub UNet class gets a no-op `from_pretrained`) and ITS `train_util.text_tokenize /
    text_encode / encode_prompts` (train_util.py:60-87) against this package's loader and glue on the same folder:
    SD1 (12 layers) and SD2 (24 layers, penultimate layer by default), `clip_skip` None / 1 / 2 / 3 -- same layer count, same
    weights, identical token ids and identical embeddings."""
    from leco_amd import model_util, train_util
    rmu = importlib.import_module("model_util")
    assert rmu.__file__.startswith(REF)
    folder = _clip_folder(tmp_path, 24 if v2 else 12)
    monkeypatch.setattr(rmu, "TOKENIZER_V1_MODEL_NAME", folder)
    monkeypatch.setattr(rmu, "TOKENIZER_V2_MODEL_NAME", folder)
    monkeypatch.setattr(rmu.UNet2DConditionModel, "from_pretrained", classmethod(lambda cls, *a, **k: None), raising=False)
    r_tok, r_enc, _ = rmu.load_diffusers_model(folder, v2=v2, clip_skip=clip_skip)
    # (our loader also opens unet/: give it the smallest valid one)
    import json
    from safetensors.torch import save_file
    from leco_amd.unet import UNet2DConditionModel
    cfg = model_util.tiny_config()
    os.makedirs(os.path.join(folder, "unet"))
    json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.__dict__.items()},
              open(os.path.join(folder, "unet", "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in model_util.init_synthetic_(UNet2DConditionModel(cfg), 3).state_dict().items()},
              os.path.join(folder, "unet", "diffusion_pytorch_model.safetensors"))
    o_tok, o_enc, _ = model_util.load_diffusers_model(folder, v2=v2, clip_skip=clip_skip)
    full = 24 if v2 else 12
    expect = full - (clip_skip - 1) if clip_skip is not None else (23 if v2 else 12)       # model_util.py:46,60
    assert r_enc.config.num_hidden_layers == o_enc.config.num_hidden_layers == expect
    rs, os_ = r_enc.state_dict(), o_enc.state_dict()
    assert list(rs) == list(os_) and all(torch.equal(rs[k], os_[k]) for k in rs)
    rt, ot = ref["train_util"], train_util
    tok_r, tok_o = rt.text_tokenize(r_tok, PROMPTS), ot.text_tokenize(o_tok, PROMPTS)
    assert tok_r.shape == (len(PROMPTS), 77) and torch.equal(tok_r, tok_o)
    with torch.no_grad():
        assert torch.equal(rt.text_encode(r_enc, tok_r), ot.text_encode(o_enc, tok_o))
        e_r, e_o = rt.encode_prompts(r_tok, r_enc, PROMPTS), ot.encode_prompts(o_tok, o_enc, PROMPTS)
    assert e_r.shape == (len(PROMPTS), 77, 64) and torch.equal(e_r, e_o)
    assert not torch.equal(e_o[0], e_o[1]) and torch.isfinite(e_o).all()


@pytest.mark.parametrize("n_img", [1, 2])
def test_prompt_front_end_xl_glue_equals_the_reference(ref, tmp_path, n_img):
    """N2, XL: the reference's `text_encode_xl` / `encode_prompts_xl` (train_util.py:90-130: penultimate hidden state of both
    encoders concatenated, pooled output of the second, `num_images_per_prompt` repeats) against this package's on real
    `CLIPTextModel` + `CLIPTextModelWithProjection` objects loaded by `load_models_xl` from a folder."""
    import json
    from safetensors.torch import save_file
    from leco_amd import model_util, train_util
    from leco_amd.unet import UNet2DConditionModel
    folder = _clip_folder(tmp_path, 4, with_projection_2=True)
    cfg = model_util.tiny_xl_config()
    os.makedirs(os.path.join(folder, "unet"))
    json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.__dict__.items()},
              open(os.path.join(folder, "unet", "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in model_util.init_synthetic_(UNet2DConditionModel(cfg), 3).state_dict().items()},
              os.path.join(folder, "unet", "diffusion_pytorch_model.safetensors"))
    toks, encs, _, _ = model_util.load_models_xl(folder, "ddim")
    rt, ot = ref["train_util"], train_util
    with torch.no_grad():
        for tok, enc in zip(toks, encs):
            ids = rt.text_tokenize(tok, PROMPTS)
            a, b = rt.text_encode_xl(enc, ids, n_img), ot.text_encode_xl(enc, ids, n_img)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        (e_r, p_r), (e_o, p_o) = rt.encode_prompts_xl(toks, encs, PROMPTS, n_img), ot.encode_prompts_xl(toks, encs, PROMPTS, n_img)
    assert e_r.shape == (len(PROMPTS) * n_img, 77, 64 + 128) and p_r.shape == (len(PROMPTS) * n_img, 96)
    assert torch.equal(e_r, e_o) and torch.equal(p_r, p_o)
