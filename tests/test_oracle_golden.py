"""CPU: the oracle (oracle/dit_ref.py) against the golden vectors produced by the imported
reference (tools/make_golden.py).  fp32 vs fp32, so tolerances are tight (accumulation order only)."""
import torch

from hunyuanvideo_efficiency_amd import synthetic as syn
from oracle import dit_ref as R

P = R.FP32
TOL = dict(rtol=2e-5, atol=2e-5)


def close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    torch.testing.assert_close(a.float(), b.float(), **tol)


def test_rmsnorm(golden):
    g = golden("dit_rmsnorm")
    close(R.rms_norm(g["x"], g["w"], P), g["y"])


def test_rope_tables_and_apply(golden):
    g = golden("dit_rope")
    cos, sin = R.rope_tables(g["sizes"].tolist(), [16, 56, 56], 256.0)
    close(cos, g["cos"], rtol=0, atol=1e-6)
    close(sin, g["sin"], rtol=0, atol=1e-6)
    # spot value of SURVEY.md 8a (a6): token 81 = (t0,h1,w1): cols 0-15 = 1, col 16 = cos(1)
    tok = 1 * 8 + 1
    assert torch.all(cos[tok, :16] == 1.0) and abs(float(cos[tok, 16]) - 0.540302) < 1e-6
    close(R.apply_rope(g["q"], g["cos"], g["sin"], P), g["qo"])
    close(R.apply_rope(g["k"], g["cos"], g["sin"], P), g["ko"])
    # rotate_half([0..7]) = [-1,0,-3,2,-5,4,-7,6]
    assert g["rot8"].flatten().tolist() == [-1, 0, -3, 2, -5, 4, -7, 6]
    g2 = golden("dit_rope_tables2")
    c2, s2 = R.rope_tables(g2["sizes"].tolist(), [16, 56, 56], 256.0)
    close(c2, g2["cos"], rtol=0, atol=1e-6)
    close(s2, g2["sin"], rtol=0, atol=1e-6)


def test_modulate_gate(golden):
    g = golden("dit_modulate")
    close(R.ln_modulate(g["x"], g["shift"], g["scale"], P), g["ln_mod"])
    ln = R.ln_modulate(g["x"], None, None, P)
    close(R.gate_residual(g["x"], ln, g["gate"], P), g["gated"])


def test_timestep_embedding(golden):
    g = golden("dit_timestep_embedding")
    e = R.timestep_embedding(g["t"])
    close(e, g["emb"], rtol=0, atol=2e-4)   # cos/sin of arguments up to 6016 rad: fp32 range reduction
    assert abs(float(e[0, 0]) - 0.562379) < 1e-5 and abs(float(e[0, 1]) - 0.789627) < 2e-4


def test_mlp_gelu_tanh(golden):
    g = golden("dit_mlp")
    w = {k: syn.synth_param("g.mlp." + k, shp, 1) for k, shp in
         {"fc1.weight": (1024, 256), "fc1.bias": (1024,), "fc2.weight": (256, 1024), "fc2.bias": (256,)}.items()}
    h = R.gelu_tanh(P.linear(g["x"], w["fc1.weight"], w["fc1.bias"]), P)
    close(P.linear(h, w["fc2.weight"], w["fc2.bias"]), g["y"])


def test_cu_seqlens(golden):
    g = golden("dit_cu_seqlens")
    cu = R.cu_seqlens(g["text_mask"], int(g["img_len"]))
    assert cu.dtype == torch.int32 and cu.tolist() == g["cu"].tolist() == [0, 331, 352, 704, 704]


def test_scheduler(golden):
    g = golden("dit_scheduler")
    for n in (30, 50):
        s = R.flow_sigmas(n, 7.0)
        close(s, g[f"sigmas{n}"], rtol=0, atol=1e-7)
        close(R.flow_timesteps(s), g[f"timesteps{n}"], rtol=0, atol=1e-4)
    s = R.flow_sigmas(50, 7.0)
    # sigma0 = 1, sigma1 = 0.997093, sigma49 = 0.125, sigma50 = 0
    assert float(s[0]) == 1.0 and abs(float(s[1]) - 0.997093) < 1e-6 and abs(float(s[49]) - 0.125) < 1e-6 and float(s[50]) == 0
    cur = g["traj"][0]
    for i in range(3):
        cur = R.euler_step(cur, g[f"v{i}"], s, i)
        assert cur.dtype == torch.float32
        close(cur, g["traj"][i + 1], rtol=0, atol=1e-7)


def test_blocks(golden):
    g = golden("dit_blocks")
    cfg = syn.tiny_config()
    sd = syn.synth_dit_state_dict(cfg, seed=0)
    cos, sin = R.rope_tables([5, 8, 8], cfg.rope_dim_list, 256.0)
    io, to = R.double_block(sd, "double_blocks.0.", g["img"], g["txt"], g["vec"], g["cu"], cos, sin, cfg.heads_num, P)
    close(io, g["img_out"], rtol=1e-4, atol=1e-4)
    close(to, g["txt_out"], rtol=1e-4, atol=1e-4)
    xs = torch.cat([g["img"], g["txt"]], 1)
    so = R.single_block(sd, "single_blocks.0.", xs, g["vec"], g["txt"].shape[1], g["cu"], cos, sin, cfg.heads_num, P)
    close(so, g["single_out"], rtol=1e-4, atol=1e-4)


def _tiny_inputs(g):
    cfg = syn.tiny_config()
    T, H, W = g["latent_thw"].tolist()
    x, ts, tm, ts2 = syn.synth_dit_inputs(cfg, (T, H, W), g["text_states"].shape[1], int(g["text_mask"].sum()), seed=0)
    # the fixture's inputs are the synthetic generator's: data and generator pin each other
    assert torch.equal(x, g["x"]) and torch.equal(ts, g["text_states"]) and torch.equal(tm, g["text_mask"])
    assert torch.equal(ts2, g["text_states_2"])
    cos, sin = R.rope_tables([T, H // 2, W // 2], cfg.rope_dim_list, 256.0)
    return cfg, syn.synth_dit_state_dict(cfg, seed=0), cos, sin


def test_tiny_forward(golden):
    g = golden("dit_tiny_forward")
    cfg, sd, cos, sin = _tiny_inputs(g)
    taps = {}
    out = R.dit_forward(sd, cfg, g["x"], g["t"], g["text_states"], g["text_mask"], g["text_states_2"], cos, sin,
                        g["guidance"], P, taps=taps)
    close(taps["img0"], g["img0"], rtol=1e-4, atol=1e-4)
    close(taps["txt0"], g["txt0"], rtol=1e-4, atol=1e-4)
    close(taps["img_d"], g["img_d"], rtol=1e-4, atol=2e-4)
    close(taps["txt_d"], g["txt_d"], rtol=1e-4, atol=2e-4)
    close(taps["x_s"], g["x_s"], rtol=1e-4, atol=2e-4)
    assert out.shape == g["out"].shape == (1, 16, 5, 16, 16)
    close(out, g["out"], rtol=1e-4, atol=2e-4)


def test_tiny_denoise_loop(golden):
    g0 = golden("dit_tiny_forward")
    g = golden("dit_tiny_denoise3")
    cfg, sd, cos, sin = _tiny_inputs(g0)
    lat, preds = R.denoise_loop(sd, cfg, g0["x"], 3, g0["text_states"], g0["text_mask"], g0["text_states_2"],
                                cos, sin, guidance_scale=6.0, shift=7.0, p=P)
    for i in range(3):
        close(preds[i], g["preds"][i], rtol=1e-4, atol=3e-4)
    close(lat, g["final_latents"], rtol=1e-4, atol=3e-4)


def test_bf16_emulation_stays_near_fp32(golden):
    """The bf16-emulated contract (what the GPU kernels are compared with) drifts from fp32 only by
    rounding: bound it so a wrong cast placement shows up here."""
    g = golden("dit_tiny_forward")
    cfg, sd, cos, sin = _tiny_inputs(g)
    out = R.dit_forward(sd, cfg, g["x"], g["t"], g["text_states"], g["text_mask"], g["text_states_2"], cos, sin,
                        g["guidance"], R.Prec(True))
    err = (out - g["out"]).abs().max() / g["out"].abs().max()
    assert err < 3e-2, float(err)


def test_fp8_weight_path(golden):
    """fp8_optimization.py: scale, e4m3fn bits, dequantised weight and Linear output of the reference vs the oracle and the
    host-side quantiser of the package."""
    g = golden("dit_fp8")
    assert float(g["maxval"]) == R.fp8_maxval() == 448.0
    scale = torch.max(torch.abs(g["w"].flatten())) / 448.0
    close(scale, g["scale"], rtol=0, atol=0)
    w8 = R.fp8_quant_dequant(g["w"], scale).to(torch.float8_e4m3fn)
    assert torch.equal(w8.view(torch.uint8), g["w8_bits"])
    close(w8.float() * scale, g["w_dequant"], rtol=0, atol=0)
    close(R.fp8_linear(g["x"], g["w"], g["b"]), g["y_fly"], rtol=1e-5, atol=1e-5)
    from hunyuanvideo_efficiency_amd.modules import fp8_optimization as F8
    q8, s = F8.quantize_weight(g["w"])
    assert torch.equal(q8.view(torch.uint8), g["w8_bits"]) and float(s) == float(g["scale"]) and F8.get_fp_maxval() == 448.0
