"""Measurement plumbing on CPU: the launch-shape tuner's table / keys / candidates, `leco_gemm_describe` (dry run of the
dispatcher: no device needed) and bench.py's attribution of plan launches to kernel-trace rows."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leco_amd import hip, tune  # noqa: E402

bf = torch.bfloat16


def _args(m, n, k, **kw):
    x = torch.zeros(8, 8, dtype=bf)       # addresses only: describe / keys never touch the memory
    return hip.gemm_args(x, x, x, m=m, n=n, k=k, **kw)


def test_describe_names_the_instantiation_the_dispatcher_would_launch():
    hip._use_library(hip.LIB_PATH) if os.path.exists(hip.LIB_PATH) else None
    ws = torch.zeros(32 * 1024 * 1024 // 4)
    wsp, wsb = ws.data_ptr(), ws.numel() * 4
    # 3x3 / stride-1 convolutions: the patch-staged kernel, launch shape from its cost model -- the level-0 conv fills the
    # chip in one round of 128x160 tiles; deep K / small M is split over whole channel chunks
    g0 = _args(16384, 320, 2880, a_mode=hip.A_CONV3_S1, conv=(4, 64, 64, 64, 64))
    assert hip.gemm_describe(g0, 0, 0, wsp, wsb) == "conv_patch_kernel<128, 160, 4, false> grid=256 split=1"
    d = hip.gemm_describe(_args(1024, 1280, 11520, a_mode=hip.A_CONV3_S1, conv=(4, 16, 16, 16, 16)), 0, 0, wsp, wsb)
    assert d.startswith("conv_patch_kernel<128, 128, 4, false> grid=80 split=3"), d
    # tile = -1 pins the implicit-GEMM family (what other gathers, and LoRA-carrying convs, run)
    assert hip.gemm_describe(g0, -1, 0, wsp, wsb).startswith("gemm_kernel<128, 160, true, 4, 4, 0> grid=256 split=1")
    d = hip.gemm_describe(_args(1024, 1280, 11520, a_mode=hip.A_CONV3_S1, conv=(4, 16, 16, 16, 16)), -1, 0, wsp, wsb)
    assert d.startswith("gemm_kernel<256, 128, true, 3, 4, 0>") and "split=6" in d
    # large plain 128x128 grids run as 4-wave workgroups (two per CU); explicit tile ids pin either form
    g = _args(16384, 2560, 320)
    assert "gemm_kernel<128, 128, false, 2, 2, 0>" in hip.gemm_describe(g, 0, 0, wsp, wsb)
    assert "gemm_kernel<128, 128, false, 4, 4, 0>" in hip.gemm_describe(g, 6, 1)
    assert "gemm_kernel<128, 128, false, 2, 2, 0>" in hip.gemm_describe(g, 5, 1)
    # fused LoRA down-projection on a deep-K small-M shape: the unsplit 64x64 grid keeps it fused (one kernel)
    x = torch.zeros(8, 8, dtype=bf)
    g = _args(1024, 1280, 5120, w_ext=x, ext_k=32, t_w=x, t_rows=16, t_out=x)
    assert hip.gemm_describe(g, 0, 0, wsp, wsb) == "gemm_kernel<64, 64, false, 4, 2, 1> grid=320 split=1"


def test_tuner_keys_candidates_and_table():
    g = _args(4096, 640, 5760, a_mode=hip.A_CONV3_S1, conv=(4, 32, 32, 32, 32), residual=torch.zeros(8, dtype=bf),
              bias=torch.zeros(8))
    key = tune.shape_key(g)
    assert key == "m4096n640k5760a1c4x32x32<32x32e0rbA0"
    cands = tune.candidates(g, has_ws=True)
    assert (0, 0) in cands and (2, 2) in cands and (4, 1) in cands and all(t != 6 for t, _ in cands)   # 6: plain only
    assert (7, 1) in cands and (8, 2) in cands and (9, 4) in cands                                     # patch-staged variants
    assert tune.shape_key(g, has_ws=False) == key + "W0"
    geglu = _args(16384, 2560, 320, act=hip.ACT_GEGLU)
    assert {t for t, _ in tune.candidates(geglu, has_ws=True)} == {0, 1, 4, 5, 6}                     # 128-column tiles
    tab = json.load(open(tune.TABLE_PATH))
    assert len(tab) > 100 and all(len(v) == 2 for v in tab.values())
    # without a GPU (or on the emulator) the tuner never measures and defers to the C heuristic
    assert tune.choose(g, None) == (0, 0)


def test_bench_attributes_launches_to_kernel_trace_rows(tmp_path, monkeypatch):
    import bench
    top = bench._profile_top_row()
    assert top and top["name"].startswith(("gemm_kernel<", "conv_patch_kernel<")) and top["share_pct"] > 5
    # counter summaries: kernel names contain commas; the row is found by its full name
    row = bench._pmc_row(bench.PMC_MFMA, top["name"])
    assert row and row["calls"] > 0 and 0.05 < row["SQ_VALU_MFMA_BUSY_CYCLES"] / (row["GRBM_GUI_ACTIVE"] / 8 * 1024) < 1
    assert bench._pmc_row(bench.PMC_FETCH, top["name"])["FETCH_SIZE"] > 0
    assert bench.step_flops(2, 25) == 2 * 2 * 0.8033e12 * (25 + 5 + 0.157)


def test_bench_dominant_kernel_accounting_on_a_tiny_step(monkeypatch):
    """bench.dominant_kernel_roofline over the plans of a real (tiny, emulated) step, with the per-launch clock replaced by a
    FLOP-proportional fake: every launch of the step is attributed to a kernel instantiation, the dominant one carries
    launches / FLOPs / algorithmic bytes per launch and the runners-up ride along -- the accounting the GPU run relies on."""
    import contextlib
    import io
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    import bench
    from leco_amd import hip, model_util, prompt_util
    from leco_amd.lora import LoRANetwork
    from leco_amd.scheduler import create_noise_scheduler
    from leco_amd.train import FusedStep
    from leco_amd.unet import UNet2DConditionModel
    hip._use_library(build_emu.build())
    try:
        _dominant_accounting(bench, monkeypatch)
    finally:
        hip._use_library(hip.LIB_PATH)      # the other tests of this file dry-run the real dispatcher
        hip._lib_path = hip.LIB_PATH


def _dominant_accounting(bench, monkeypatch):
    import contextlib
    import io
    from leco_amd import model_util, prompt_util
    from leco_amd.lora import LoRANetwork
    from leco_amd.scheduler import create_noise_scheduler
    from leco_amd.train import FusedStep
    from leco_amd.unet import UNet2DConditionModel
    m = model_util.init_synthetic_(UNet2DConditionModel(model_util.tiny_config()), 1234).to(torch.bfloat16)
    m.requires_grad_(False)
    m.use_graphs = False
    with contextlib.redirect_stdout(io.StringIO()):
        net = LoRANetwork(m, rank=4)
    eg = torch.Generator().manual_seed(5)
    emb = [torch.randn(1, 77, 64, generator=eg) for _ in range(4)]
    s_ = prompt_util.PromptSettings(target="t", positive="p", neutral="n", unconditional="u", guidance_scale=2.0, batch_size=1,
                                    resolution=128)
    pair = prompt_util.PromptEmbedsPair(torch.nn.MSELoss(), emb[0], emb[1], emb[2], emb[3], s_)
    fs = FusedStep(m, net, create_noise_scheduler("ddim"), 10, lr=1e-3)
    fs.step(pair, 1, torch.randn(1, 4, 16, 16, generator=eg))
    st = fs._state[(1, 16, 16)]
    launches = bench.step_launches(st, 3.0)
    assert len(launches) > 100 and all(len(bench._launch_identity(op)) == 4 for op, _, _ in launches[:50])
    assert {which for _, _, which in launches} == {"ctx_on", "denoise", "fwd_off", "fwd_on", "bwd"}
    monkeypatch.setattr(bench, "_time_launch_us", lambda op, reps=8: 5.0 + bench._launch_identity(op)[2] / 1e9)
    monkeypatch.setattr(bench, "PROFILE_STATS", os.path.join(ROOT, "profiles", "does_not_exist.txt"))
    d = bench.dominant_kernel_roofline(st, 3.0)
    assert d["name"] == d["live_top"] and d["profile_top_row"] is None and not d["agrees_with_profile"]
    assert d["launches_per_step"] > 0 and d["flops_per_launch"] > 0 and d["algorithmic_bytes_per_launch"] > 0
    assert 0 < d["share_of_step_kernel_time"] < 1 and len(d["next_kernels"]) == 3
    assert all(k["name"] != d["name"] and "frac" in k for k in d["next_kernels"])
    assert d["traffic"] is None or d["traffic"] > 0
    # the isolated GPU time of the step's launch lists (bench.py's gpu_busy_frac): fixed lists + per denoising pass, and the
    # live figure IS the headline fraction (the committed trace only rides along as frac_trace)
    iso = d["isolated_us"]
    n_den = sum(1 for _, _, which in launches if which == "denoise")
    assert iso["launches"]["denoise"] == n_den and iso["per_denoise_pass"] >= 5.0 * n_den
    assert abs(iso["fixed"] + iso["per_denoise_pass"] - sum(iso["by_list"].values())) < 1e-6 * iso["fixed"]
    assert d["timing_source"].startswith("live") and "frac_trace" not in d
    # a committed trace on the running kernel sources names the dominant kernel of the HEADLINE workload only: another
    # configuration's line (BASELINE configs 3 - 5) ranks its kernels live and quotes no trace / counter figure
    import tempfile
    runner_up = d["next_kernels"][0]["name"]
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as fh:
        fh.write(f"# csrc_sha1={bench.kernel_sources_hash()}\n     %    calls   total_ms    avg_us   min_us    max_us  kernel\n"
                 f" 50.00      100     1.00      10.0      9.0      11.0  {runner_up}\n")
    monkeypatch.setattr(bench, "PROFILE_STATS", fh.name)
    try:
        h = bench.dominant_kernel_roofline(st, 3.0, headline_workload=True)
        assert h["name"] == runner_up and h["agrees_with_profile"] and h["us_per_launch_trace"] == 10.0 and h["live_top"] == d["name"]
        o = bench.dominant_kernel_roofline(st, 3.0, headline_workload=False)
        assert o["name"] == d["name"] and o["profile_top_row"] is None and o["agrees_with_profile"] is None and "frac_trace" not in o
        assert o["traffic"] is None and "headline workload" in o["traffic_note"]
    finally:
        os.unlink(fh.name)


def test_gpu_telemetry_degrades_to_fields_and_summarises():
    """tools/gpu_telemetry.py (bench.py's clock / power / throttle sampler) must never raise: without a GPU every reading is
    an (almost) empty dict with the reason as a field; the summary / delta helpers are plain arithmetic."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gpu_telemetry import Telemetry
    t = Telemetry(0)
    snap = t.snapshot()
    assert isinstance(snap, dict) and snap["source"] in ("amdsmi", "sysfs", "none")
    t.start(50.0)
    import time
    time.sleep(0.1)
    ss = t.stop()
    assert len(ss) >= 2 and all("t" in s for s in ss)
    fake = [{"t": 1.0, "sclk_mhz": 2100.0, "power_w": 700}, {"t": 2.0, "sclk_mhz": 1900.0, "power_w": 900, "throttle_status": 4},
            {"t": 9.0, "sclk_mhz": 100.0}]
    s = Telemetry.summarize(fake, 0.5, 2.5)
    assert s["n"] == 2 and s["sclk_mhz"] == {"min": 1900.0, "median": 2100.0, "max": 2100.0, "first": 2100.0, "last": 1900.0}
    assert s["throttle_flags_seen"] == ["4"]
    d = Telemetry.delta({"ppt_residency_acc": 10, "energy_accumulator": 5}, {"ppt_residency_acc": 25, "energy_accumulator": 9})
    assert d == {"ppt_residency_acc": 15, "energy_accumulator": 4}
