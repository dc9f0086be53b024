"""Host-side logic on CPU: the C-ABI library loads and exports every symbol include/leco_hip.h declares
(no compute calls here), config / prompt schemas, the DDIM scheduler against the oracle, LoRA save format."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "leco_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(leco_[a-z0-9_]+)\s*\(", src)))


def test_c_abi_library_builds_loads_and_exports_every_declared_symbol(tmp_path):
    import __graft_entry__
    __graft_entry__.build()           # hipcc cross-compiles gfx950 without a GPU
    from leco_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/leco_hip.h but not exported"
    lib.leco_version.restype = ctypes.c_int
    assert lib.leco_version() >= 100
    # ... and the other way round: nothing is exported that the header does not declare (the boundary IS the header)
    nm = subprocess.run(["nm", "-D", "--defined-only", hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if l.split() and l.split()[-1].startswith("leco_")}
    assert exported and exported <= set(syms), sorted(exported - set(syms))
    # the device code object really targets gfx950
    # (llvm-objdump --offloading unbundles the code objects next to its INPUT: give it a scratch copy)
    import shutil
    scratch = shutil.copy(hip.LIB_PATH, tmp_path / "libleco_hip.so")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(scratch)], capture_output=True,
                         text=True, cwd=tmp_path).stdout
    assert "gfx950" in out


def test_async_lds_reads_are_never_touched_in_flight():
    """ISA audit (tools/audit_async_lds.py): no instruction of any GEMM instantiation names the destination of an
    asynchronous asm ds_read before the s_waitcnt that covers it -- the compiler-inserted register copies on loop
    edges that produced the round-1 full-size NaN."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_async_lds
    report = audit_async_lds.audit_source(os.path.join(ROOT, "leco_amd", "csrc", "gemm.hip"))
    assert len(report) >= 15
    patch = audit_async_lds.audit_source(os.path.join(ROOT, "leco_amd", "csrc", "conv_patch.hip"))
    assert len(patch) >= 4
    report = report + patch
    bad = {audit_async_lds.pretty(name): list(v.items())[:3] for name, _, v in report if v}
    assert not bad, bad


def test_product_path_fails_loudly_without_the_extension(monkeypatch, tmp_path):
    from leco_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip.lib()


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "leco_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_config_defaults_and_coercions(tmp_path):
    from leco_amd import config_util
    p = tmp_path / "c.yaml"
    p.write_text('prompts_file: "p.yaml"\npretrained_model:\n  name_or_path: "synthetic:tiny"\nnetwork:\n  rank: 8\n'
                 'train:\n  lr: 1e-4\n  batch_size: 2\n')
    c = config_util.load_config_from_yaml(str(p))
    assert c.train.lr == 1e-4 and c.train.max_denoising_steps == 50 and c.train.precision == "bfloat16"
    assert c.save.per_steps == 200 and c.save.name == "untitled" and c.logging.use_wandb is False
    assert c.network.type == "lierla" and c.network.alpha == 1.0 and c.network.training_method == "full"
    assert config_util.parse_precision("bf16") is torch.bfloat16
    with pytest.raises(ValueError):
        config_util.parse_precision("int8")


def test_prompt_settings_defaults(tmp_path):
    from leco_amd import prompt_util
    s = prompt_util.PromptSettings(target="van gogh")
    assert (s.positive, s.unconditional, s.neutral, s.action, s.batch_size, s.resolution) == ("van gogh", "", "", "erase", 1, 512)
    with pytest.raises(Exception):
        prompt_util.PromptSettings(positive="x")
    p = tmp_path / "empty.yaml"
    p.write_text("[]\n")
    with pytest.raises(ValueError):
        prompt_util.load_prompts_from_yaml(str(p))


@pytest.mark.parametrize("pt", ["epsilon", "v_prediction"])
def test_ddim_scheduler_matches_oracle(pt):
    from leco_amd.scheduler import DDIMScheduler, create_noise_scheduler
    from oracle.ddim_ref import DDIMSchedulerRef
    a, b = DDIMScheduler(prediction_type=pt), DDIMSchedulerRef(prediction_type=pt)
    for n in (50, 30, 1000):
        a.set_timesteps(n)
        b.set_timesteps(n)
        assert torch.equal(a.timesteps, b.timesteps)
    a.set_timesteps(50)
    b.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    x, e = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    for t in a.timesteps:
        assert torch.allclose(a.step(e, t, x).prev_sample, b.step(e, t, x).prev_sample, rtol=1e-5, atol=1e-6)
    tab = a.coef_table()
    assert tab.shape == (50, 2)
    assert torch.allclose(tab[3, 0] * x + tab[3, 1] * e, b.step(e, a.timesteps[3], x).prev_sample, rtol=1e-5, atol=1e-6)
    assert a.init_noise_sigma == 1.0 and a.scale_model_input(x, 5) is x
    for name in ("ddpm", "lms", "euler_a"):      # model_util.py:247-274: every accepted name builds
        assert create_noise_scheduler(name, prediction_type=pt).prediction_type == pt
    with pytest.raises(ValueError):
        create_noise_scheduler("dpm++")


def test_lora_save_format_round_trip(tmp_path):
    import contextlib
    import io
    from safetensors.torch import load_file
    from leco_amd import model_util
    from leco_amd.lora import LoRANetwork
    from leco_amd.unet import UNet2DConditionModel
    with torch.device("meta"):
        m = UNet2DConditionModel(model_util.tiny_config())
    with contextlib.redirect_stdout(io.StringIO()):
        net = LoRANetwork(m, rank=4, alpha=1.0)
    f = tmp_path / "x_last.safetensors"
    net.save_weights(f, dtype=torch.bfloat16)
    sd = load_file(str(f))
    want = [l.split(" ")[0] for l in open(os.path.join(ROOT, "tests", "golden", "tiny_lora_keys.txt"))]
    assert sorted(sd.keys()) == sorted(want)
    k = "lora_unet_down_blocks_0_attentions_0_proj_in"
    assert sd[k + ".lora_down.weight"].shape == (4, 64, 1, 1) and sd[k + ".lora_up.weight"].shape == (64, 4, 1, 1)
    assert sd[k + ".alpha"].ndim == 0 and sd[k + ".alpha"].dtype == torch.bfloat16
    assert sd["lora_unet_mid_block_attentions_0_transformer_blocks_0_ff_net_0_proj.lora_down.weight"].shape == (4, 128)
    assert (sd[k + ".lora_up.weight"] == 0).all()            # lora.py:92
    # parameters are views into one flat slab
    p0 = net.unet_loras[0].lora_down.weight
    assert p0.data_ptr() == net.slab.data_ptr() and net.numel == sum(p.numel() for p in net.parameters())


# ---- single-file (LDM layout) checkpoints: model_util.py:75-101 ------------------------------------------
def test_ldm_key_layout_known_names_and_counts():
    """Pins the LDM <-> diffusers map against public facts: tensor counts of the released UNets (686 for SD1.x /
    SD2.x, 1680 for SDXL) and well-known parameter names of the CompVis / SGM layouts."""
    from leco_amd import ckpt_convert as cc
    from leco_amd.unet import UNet2DConditionModel, sd15_config, sd21_config, sdxl_config
    known = {
        "sd15": (sd15_config(), 686, ["input_blocks.0.0.weight", "input_blocks.3.0.op.weight", "input_blocks.11.0.in_layers.2.weight",
                                      "input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight", "middle_block.1.proj_in.weight",
                                      "output_blocks.2.1.conv.weight", "output_blocks.5.2.conv.weight",
                                      "output_blocks.11.1.transformer_blocks.0.ff.net.0.proj.weight", "out.2.weight",
                                      "time_embed.2.bias", "input_blocks.4.0.skip_connection.weight"]),
        "sd21": (sd21_config(), 686, ["input_blocks.8.1.proj_out.weight", "output_blocks.8.2.conv.weight"]),
        "sdxl": (sdxl_config(), 1680, ["label_emb.0.0.weight", "input_blocks.4.1.transformer_blocks.1.attn1.to_q.weight",
                                       "input_blocks.8.1.transformer_blocks.9.ff.net.2.weight", "output_blocks.2.2.conv.weight",
                                       "output_blocks.5.2.conv.weight", "output_blocks.8.0.skip_connection.weight"]),
    }
    for name, (cfg, count, names) in known.items():
        with torch.device("meta"):
            sd = UNet2DConditionModel(cfg).state_dict()
        assert len(sd) == count, name
        ldm = cc.diffusers_unet_to_ldm(sd, cfg)
        for n in names:
            assert cc.UNET_PREFIX + n in ldm, (name, n)
        assert set(cc.convert_ldm_unet(ldm, cfg)) == set(sd)
        assert cc.detect_unet_config(ldm) == cfg, name
    # shapes agree where the layouts are known to differ in rank: SD1.x proj_in is a 1x1 conv, SD2.x a Linear
    assert len(known and ldm) > 0


@pytest.mark.parametrize("arch", ["sd15", "sd21", "sdxl"])
def test_ldm_converter_against_the_public_checkpoint_key_list(arch):
    """N1 pinned to something that is NOT `ckpt_convert`: tests/golden/ldm_unet_keys.json holds every UNet tensor name +
    shape of the released single-file checkpoints (686 / 686 / 1680 tensors, 859 520 964 / 865 910 724 / 2 567 463 684
    parameters -- the public totals), produced by oracle/ldm_unet_keys.py, a restatement of the CONSTRUCTOR of the public
    LDM / SGM `UNetModel` that shares no code with the converter (whose map is derived from the diffusers block structure).
    `detect_unet_config` must recognise the architecture from those shapes, `convert_ldm_unet` must map the full list
    ONE-TO-ONE onto the state-dict keys of the fp32 oracle UNet (oracle/unet_ref.py, the diffusers-0.20 layout) and of the
    HIP-backed module, every shape matching."""
    import json
    from leco_amd import ckpt_convert as cc
    from leco_amd import unet as U
    from oracle import ldm_unet_keys as L
    from oracle import unet_ref as R
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "ldm_unet_keys.json")))[arch]
    assert fixture == L.ldm_unet_keys(arch)                       # the committed generator made the committed list
    n_pub, p_pub = L.PUBLIC_COUNTS[arch]
    assert len(fixture) == n_pub and sum(L.numel(s) for s in fixture.values()) == p_pub
    sd = {k: torch.empty(shape, device="meta") for k, shape in fixture.items()}
    # what else a released file holds must be ignored: VAE, text encoder, EMA bookkeeping
    sd["first_stage_model.decoder.conv_in.weight"] = torch.empty(512, 4, 3, 3, device="meta")
    sd["model_ema.decay"] = torch.empty((), device="meta")
    cfg = cc.detect_unet_config(sd)
    assert cfg == getattr(U, arch + "_config")()
    conv = cc.convert_ldm_unet(sd, cfg)
    assert len(conv) == len(fixture)                               # no two checkpoint tensors land on one name
    with torch.device("meta"):
        want = {"oracle": R.UNet2DConditionModel(getattr(R, arch + "_config")()).state_dict(),
                "hip module": U.UNet2DConditionModel(cfg).state_dict()}
    for who, w in want.items():
        assert set(conv) == set(w), (who, sorted(set(conv) ^ set(w))[:6])
        bad = [k for k in w if tuple(conv[k].shape) != tuple(w[k].shape)]
        assert not bad, (who, bad[:6])
    assert set(cc.diffusers_unet_to_ldm(conv, cfg)) == set(fixture)


def test_single_file_checkpoint_round_trip(tmp_path):
    from safetensors.torch import save_file
    from leco_amd import ckpt_convert as cc, model_util
    for kind in ("tiny", "tiny_xl"):
        cfg = model_util.tiny_xl_config() if kind == "tiny_xl" else model_util.SYNTHETIC[kind]()
        from leco_amd.unet import UNet2DConditionModel
        src = model_util.init_synthetic_(UNet2DConditionModel(cfg), 7)
        ldm = cc.diffusers_unet_to_ldm(src.state_dict(), cfg)
        ldm["first_stage_model.decoder.conv_in.weight"] = torch.zeros(4, 4, 3, 3)   # VAE entries are ignored
        path = str(tmp_path / f"{kind}.safetensors")
        save_file({k: v.contiguous() for k, v in ldm.items()}, path)
        unet = model_util.load_unet_single_file(path)
        assert unet.cfg.block_out_channels == cfg.block_out_channels
        assert unet.cfg.down_block_types == cfg.down_block_types and unet.cfg.use_linear_projection == cfg.use_linear_projection
        assert (unet.cfg.addition_embed_type == "text_time") == (cfg.addition_embed_type == "text_time")
        a, b = src.state_dict(), unet.state_dict()
        assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    with pytest.raises(KeyError):
        bad = dict(ldm)
        bad[cc.UNET_PREFIX + "input_blocks.99.0.in_layers.0.weight"] = torch.zeros(4)
        save_file(bad, path)
        model_util.load_unet_single_file(path)


def test_open_clip_text_tower_conversion():
    """SD2.x checkpoints store the OpenCLIP text tower with a fused in_proj: q / k / v are split on load."""
    from leco_amd import ckpt_convert as cc
    d, pre = 16, "cond_stage_model.model."
    sd = {pre + "token_embedding.weight": torch.randn(10, d), pre + "positional_embedding": torch.randn(77, d),
          pre + "ln_final.weight": torch.randn(d), pre + "ln_final.bias": torch.randn(d),
          pre + "transformer.resblocks.0.attn.in_proj_weight": torch.arange(3 * d * d, dtype=torch.float32).reshape(3 * d, d),
          pre + "transformer.resblocks.0.attn.in_proj_bias": torch.arange(3 * d, dtype=torch.float32),
          pre + "transformer.resblocks.0.attn.out_proj.weight": torch.randn(d, d),
          pre + "transformer.resblocks.0.ln_1.weight": torch.randn(d), pre + "transformer.resblocks.0.mlp.c_fc.bias": torch.randn(4 * d)}
    out = cc.convert_open_clip(sd)
    base = "text_model.encoder.layers.0."
    assert torch.equal(out[base + "self_attn.k_proj.weight"], sd[pre + "transformer.resblocks.0.attn.in_proj_weight"][d:2 * d])
    assert torch.equal(out[base + "self_attn.v_proj.bias"], sd[pre + "transformer.resblocks.0.attn.in_proj_bias"][2 * d:])
    assert base + "layer_norm1.weight" in out and base + "mlp.fc1.bias" in out and base + "self_attn.out_proj.weight" in out
    assert "text_model.embeddings.position_embedding.weight" in out and "text_model.final_layer_norm.bias" in out


def test_gemm_dma_protocol_under_late_completion():
    """The emulator normally lands an LDS-DMA at issue; with LECO_EMU_DMA=late it lands only at the issuing lane's
    counted `s_waitcnt vmcnt(N)` (or a draining __syncthreads) -- the latest moment the hardware allows.  The GEMM /
    conv / attention kernels (counted waits across a raw barrier, 3-4 tiles in flight) must give the same results in
    both models; a missing or mis-counted wait shows up as stale LDS data (rel. error ~1 instead of 1e-7).  The whole
    kernel test file runs in that model (every kernel that stages through the DMA, incl. the A-stationary GEMM, the
    LayerNorm fold and the split-K consumers)."""
    env = dict(os.environ, LECO_EMU_DMA="late")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels.py"), "-q", "-x",
                        "-m", "not gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_emulator_schedules_order_the_waves_and_expose_a_missing_barrier():
    """The emulator's work-item schedules (LECO_EMU_SCHED, tests/emu/emu_runtime.cpp) on their own test program
    (tests/emu/selftest_sched.cpp): the order in which four waves pass three rendezvous under each schedule, and an LDS
    slot reused without the closing barrier -- a race the near-lockstep round-robin order cannot see and the greedy /
    random orders must."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    exe = build_emu.build_selftest()

    def run(mode):
        r = subprocess.run([exe], env=dict(os.environ, LECO_EMU_SCHED=mode, LECO_EMU_THREADS="1"),
                           capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        lines = dict(l.split(": ", 1) for l in r.stdout.strip().splitlines())
        return [int(x) for x in lines["order"].split()], lines["reuse fenced=1"], lines["reuse fenced=0"]

    assert run("rr") == ([0, 1, 2, 3] * 3, "ok", "ok")              # lockstep: blind to the missing barrier
    assert run("reverse") == ([3, 2, 1, 0] * 3, "ok", "ok")
    assert run("greedy") == ([0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3], "ok", "RACE")
    assert run("greedy_reverse") == ([3, 3, 3, 2, 2, 2, 1, 1, 1, 0, 0, 0], "ok", "RACE")
    order, fenced, unfenced = run("random")
    assert sorted(order) == sorted([0, 1, 2, 3] * 3) and order != [0, 1, 2, 3] * 3 and fenced == "ok"
    # LECO_EMU_LDS=poison: never-written dynamic LDS reads as the NaN fill, not as the previous workgroup's leftovers
    def line(key, **env):
        r = subprocess.run([exe], env=dict(os.environ, LECO_EMU_THREADS="1", **env), capture_output=True, text=True, timeout=60)
        return [l for l in r.stdout.splitlines() if l.startswith(key)][0]
    assert line("lds") == "lds: 0 1 1 1" and line("lds", LECO_EMU_LDS="poison") == "lds: 7fc0 7fc0 7fc0 7fc0"
    # a returned wave leaves the workgroup barrier (loader waves that end before the epilogue): no deadlock, under every order
    for mode in ("rr", "reverse", "greedy", "greedy_reverse", "random"):
        assert line("early", LECO_EMU_SCHED=mode) == "early: 13 12"
    # LECO_EMU_BLOCKS: workgroup dispatch order (ascending like the hardware's by default; reversed; a fixed permutation)
    assert line("blocks") == "blocks: 0 1 2 3 4 5" and line("blocks", LECO_EMU_BLOCKS="reverse") == "blocks: 5 4 3 2 1 0"
    perm = [int(x) for x in line("blocks", LECO_EMU_BLOCKS="random").split()[1:]]
    assert sorted(perm) == list(range(6)) and perm != list(range(6))


@pytest.mark.parametrize("sched,lds,blocks", [("greedy", "poison", "reverse"), ("greedy_reverse", "", "linear"),
                                              ("random", "poison", "random")])
def test_kernels_give_the_same_results_under_every_wave_schedule(sched, lds, blocks):
    """Any interleaving of a workgroup's waves between its barriers is legal on the hardware: the whole kernel test
    file must pass when one wave runs as far ahead of the others as the barriers allow (either end first) and under a
    random wave order, not only in the default near-lockstep order -- an LDS buffer refilled or reused without a
    barrier shows up here as a wrong result (see the self-test above).  Two of the three runs also start every
    workgroup on NaN-filled dynamic LDS (what the hardware leaves there is the previous workgroup's) and dispatch the
    workgroups of a launch last-first / in a scrambled order (kernels that hand data from workgroup to workgroup --
    split-K last-arriver reductions, atomically summed statistics -- must not care)."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels.py"), "-q", "-x",
                        "-m", "not gpu", "-p", "no:cacheprovider"], cwd=ROOT,
                       env=dict(os.environ, LECO_EMU_SCHED=sched, LECO_EMU_LDS=lds, LECO_EMU_BLOCKS=blocks),
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_kernel_sources_are_clean_under_address_sanitizer():
    """tools/emu_asan.py: the kernel test file with the kernel sources compiled under AddressSanitizer and the
    interpreter's allocator replaced by the sanitizer's (red zones around every CPU tensor): any global-memory access of a
    kernel outside the operand it belongs to aborts the run with the .hip source line."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    if not os.path.exists(build_emu.ASAN_RT):
        pytest.skip("no AddressSanitizer runtime in this toolchain")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emu_asan.py")], cwd=ROOT, capture_output=True,
                       text=True, timeout=1800)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr, r.stdout[-1500:] + r.stderr[-3000:]
    # and the sanitizer is live in that configuration: a deliberately short output tensor is reported
    probe = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import build_emu; "
             "from leco_amd import hip, ops; hip._use_library(build_emu.build()); "
             "x = torch.randn(1000); y = torch.empty(800, dtype=torch.bfloat16); "
             "ops.run_plan([ops.cast_f32_bf16(x, y, 1000)])"
             % (ROOT, os.path.join(ROOT, "tests", "emu")))
    env = dict(os.environ, LECO_EMU_ASAN="1", LD_PRELOAD=build_emu.ASAN_RT,
               ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    r = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "heap-buffer-overflow" in r.stderr and "cast_f32_bf16_kernel" in r.stderr, r.stderr[-3000:]


def test_no_two_workgroups_race_on_global_memory_under_thread_sanitizer():
    """tools/emu_asan.py --tsan: the emulator runs the workgroups of a launch on a pool of OS threads, so ThreadSanitizer sees
    workgroup against workgroup -- two of them touching the same global word without an atomic (work-items of ONE workgroup
    are sequential here and ordered for the sanitizer).  First the probe (64 workgroups incrementing one word with a plain
    read-modify-write: updates are lost and the race is reported; with atomicAdd neither), then the whole kernel test file:
    no report may have emulator frames on both sides."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    if not os.path.exists(build_emu.TSAN_RT):
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    exe = build_emu.build_selftest(tsan=True)
    env = dict(os.environ, LECO_EMU_THREADS="4")
    ok = subprocess.run([exe, "cross", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and ok.stdout.strip() == "cross atomic=1: 1280000" and "ThreadSanitizer" not in ok.stderr, ok.stderr[-2000:]
    bad = subprocess.run([exe, "cross", "0"], env=env, capture_output=True, text=True, timeout=120)
    assert "ThreadSanitizer: data race" in bad.stderr and "cross_block_kernel" in bad.stderr, bad.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emu_asan.py"), "--tsan"], cwd=ROOT, capture_output=True,
                       text=True, timeout=2400)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    reports = [b for b in re.split(r"={18}\n", r.stdout + r.stderr) if "WARNING: ThreadSanitizer" in b]

    def kernel_on_both_sides(b):        # the two access stacks of a report: this access, then "Previous ..." up to "Location" / "Thread"
        head = re.split(r"\n\s*(?:Location is|Thread T|Mutex M|As if)", b, maxsplit=1)[0]
        sides = re.split(r"\n\s*Previous ", head, maxsplit=1)
        return len(sides) == 2 and all("/leco_amd/csrc/" in x for x in sides)
    # not ours: torch's own worker threads; a work-item against its own scheduler (emulator state on ONE OS thread)
    ours = [b for b in reports if kernel_on_both_sides(b)]
    assert not ours, ours[0][:3000]


def _write_synthetic_clip(folder, hidden=64, layers=3):
    """A tiny but real transformers CLIP text stack on disk: tokenizer files + text_encoder/ (HF format)."""
    import json
    from transformers import CLIPTextConfig, CLIPTextModel
    chars = list("abcdefghijklmnopqrstuvwxyz")
    vocab = {c: i for i, c in enumerate(chars)}
    vocab.update({c + "</w>": len(chars) + i for i, c in enumerate(chars)})
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    os.makedirs(os.path.join(folder, "tokenizer"), exist_ok=True)
    json.dump(vocab, open(os.path.join(folder, "tokenizer", "vocab.json"), "w"))
    open(os.path.join(folder, "tokenizer", "merges.txt"), "w").write("#version: 0.2\n")
    json.dump({"model_max_length": 77, "bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>",
               "unk_token": "<|endoftext|>", "pad_token": "<|endoftext|>", "tokenizer_class": "CLIPTokenizer"},
              open(os.path.join(folder, "tokenizer", "tokenizer_config.json"), "w"))
    torch.manual_seed(0)
    te = CLIPTextModel(CLIPTextConfig(vocab_size=len(vocab), hidden_size=hidden, intermediate_size=2 * hidden,
                                      num_hidden_layers=layers, num_attention_heads=max(1, hidden // 64),
                                      max_position_embeddings=77, projection_dim=hidden,
                                      bos_token_id=len(vocab) - 2, eos_token_id=len(vocab) - 1))
    return te


def test_real_clip_front_end_from_a_diffusers_folder_and_from_a_single_file(tmp_path):
    """N1 + N2 with the installed transformers: `load_models` on (a) a diffusers-format folder (unet/, tokenizer/,
    text_encoder/; model_util.py:30-72) and (b) a single-file LDM checkpoint with the tokenizer next to it
    (model_util.py:75-101); both feed `train_util.encode_prompts` (train_lora.py:109-137) and agree."""
    import json
    from safetensors.torch import save_file
    from leco_amd import ckpt_convert as cc, model_util, train_util
    from leco_amd.unet import UNet2DConditionModel
    cfg = model_util.tiny_config()
    unet = model_util.init_synthetic_(UNet2DConditionModel(cfg), 3)
    folder = str(tmp_path / "model")
    te = _write_synthetic_clip(folder, hidden=cfg.cross_attention_dim)
    te.save_pretrained(os.path.join(folder, "text_encoder"))
    os.makedirs(os.path.join(folder, "unet"))
    json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.__dict__.items()},
              open(os.path.join(folder, "unet", "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in unet.state_dict().items()},
              os.path.join(folder, "unet", "diffusion_pytorch_model.safetensors"))
    tok, enc, u2, sched = model_util.load_models(folder, "ddim")
    a, b = unet.state_dict(), u2.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    e_folder = train_util.encode_prompts(tok, enc, ["van gogh", ""])
    assert e_folder.shape == (2, 77, cfg.cross_attention_dim) and torch.isfinite(e_folder).all()
    assert not torch.equal(e_folder[0], e_folder[1])
    # (b) the same weights as ONE file in the LDM layout: UNet under model.diffusion_model.*, CLIP under
    # cond_stage_model.transformer.text_model.* (the names SD1.x checkpoints use)
    ldm = cc.diffusers_unet_to_ldm(unet.state_dict(), cfg)
    for k, v in te.state_dict().items():
        k = k if k.startswith("text_model.") else "text_model." + k
        ldm["cond_stage_model.transformer." + k] = v
    ck = str(tmp_path / "model" / "tiny-sd.safetensors")
    save_file({k: v.contiguous() for k, v in ldm.items()}, ck)
    tok2, enc2, u3, _ = model_util.load_models(ck, "ddim")
    assert all(torch.equal(a[k], u3.state_dict()[k]) for k in a)
    e_file = train_util.encode_prompts(tok2, enc2, ["van gogh", ""])
    assert torch.allclose(e_file, e_folder, atol=1e-6)
    # clip_skip drops layers from the top (model_util.py:92-96)
    _, enc3, _ = model_util.load_checkpoint_model(ck, clip_skip=2)
    assert len([k for k in enc3.state_dict() if k.endswith("layer_norm1.weight")]) == 2


@pytest.mark.parametrize("env,select", [
    ({"LECO_GEMM_W4_MIN_BLOCKS": "1", "LECO_EMU_DMA": "late"}, "gemm"),
    ({"LECO_ATTN_QF": "1"}, "attention")])
def test_tuning_switch_variants_stay_correct(env, select):
    """Launch-shape switches (the 4-wave / two-workgroups-per-CU GEMM that large plain grids use, forced here for
    every grid; one query fragment per wave) select different kernel instantiations: each must pass the same parity
    tests, in the deferred-DMA model where DMA is involved."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels.py"), "-q", "-x",
                        "-m", "not gpu", "-k", select, "-p", "no:cacheprovider"],
                       cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_roctx_ranges_bracket_the_step_phases(monkeypatch):
    """LECO_ROCTX=1: every push has its pop and the phase names of a step appear in order (leco_amd/trace.py; the library
    call itself is replaced by a recorder -- librocprofiler-sdk-roctx needs no GPU, but the test should not depend on it)."""
    from leco_amd import trace

    class Rec:
        def __init__(self):
            self.ev = []

        def roctxRangePushA(self, s):
            self.ev.append(("push", s.decode()))
            return 0

        def roctxRangePop(self):
            self.ev.append(("pop", None))
            return 0
    rec = Rec()
    monkeypatch.setattr(trace, "_on", True)
    monkeypatch.setattr(trace, "_lib", rec)
    trace.push("denoise k=3")
    trace.pop()
    assert rec.ev == [("push", "denoise k=3"), ("pop", None)] and trace.enabled()
    monkeypatch.setattr(trace, "_on", False)
    trace.push("x")
    trace.pop()
    assert len(rec.ev) == 2                              # disabled: nothing recorded
    import inspect
    from leco_amd.train import FusedStep
    src = inspect.getsource(FusedStep.step)
    assert src.count("trace.push(") == src.count("trace.pop()") == 6
