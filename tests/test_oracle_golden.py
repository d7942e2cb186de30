"""Pin the CPU oracle against outputs of the reference's own modules (tests/golden/,
made by tools/gen_golden.py).  The reference holds no test for soccer_diffusion/ml/, so
these generated vectors are the only pins."""

import torch

from conftest import rel_err
from oracle import ddim_ref
from oracle import denoiser_ref as ref

TOL = 2e-6  # fp32 restatement vs fp32 reference: summation-order noise only


def test_step_token_int_and_float(g1):
    sd, d = g1["state_dict"], g1["config"]["d"]
    for steps, want in ((g1["steps_int"], g1["step_token_int"]), (g1["steps_float"], g1["step_token_float"])):
        got = ref.step_token(steps, sd["step_encoding.token"], d)
        assert got.shape == want.shape
        assert torch.equal(got, want)


def test_positional_table(g1):
    c = g1["config"]
    assert torch.equal(ref.positional_table(c["d"], c["T"]), g1["pe"])


def test_decoder_forward(g1):
    sd = g1["state_dict"]
    mem = torch.cat([g1["ctx"], g1["step_token_int"]], dim=1)
    assert rel_err(ref.denoiser_forward(sd, g1["x"], mem), g1["decoder_out"]) < TOL


def test_forward_with_context(g1):
    sd = g1["state_dict"]
    assert rel_err(ref.forward_with_context(sd, [g1["ctx"]], g1["x"], g1["steps_int"]), g1["eps_int"]) < TOL
    assert rel_err(ref.forward_with_context(sd, [g1["ctx"]], g1["x"], g1["steps_float"]), g1["eps_float"]) < TOL


def test_short_horizon_slices_pe(g1):
    sd = g1["state_dict"]
    got = ref.forward_with_context(sd, [g1["ctx"]], g1["x"][:, :7], g1["steps_int"])
    assert rel_err(got, g1["eps_short"]) < TOL


def test_fp64_oracle_agrees(g1):
    sd = g1["state_dict"]
    got = ref.forward_with_context(sd, [g1["ctx"]], g1["x"], g1["steps_int"], dtype=torch.float64)
    assert rel_err(got, g1["eps_int"]) < TOL


def test_full_model_encoders(g2):
    sd = g2["state_dict"]
    enc = ref.encode_input_data(sd, g2["input_data"])
    assert len(enc) == len(g2["encoded"]) == 4
    for got, want in zip(enc, g2["encoded"]):
        assert got.shape == want.shape
        assert rel_err(got, want) < TOL
    assert rel_err(ref.forward(sd, g2["input_data"], g2["x"], g2["steps"]), g2["eps"]) < TOL


def test_c2_shape_seeded_weights(g3):
    c = g3["config"]
    sd = ref.synthetic_state_dict(c["d"], c["J"], c["L"], seed=c["weight_seed"])
    checksum = torch.stack([v.double().sum() for v in sd.values()]).sum()
    assert abs(float(checksum) - float(g3["weight_checksum"])) < 1e-9, "CPU RNG stream drifted"
    assert rel_err(ref.forward_with_context(sd, [g3["ctx"]], g3["x"], g3["steps"]), g3["eps"]) < TOL


def test_train_step_loss_and_grads(g1, g2):
    """loss + per-parameter grads (dropout p=0) of the reference's training step."""
    tr = g1["train"]
    pred, loss, grads = ref.train_loss_and_grads(g1["state_dict"], g1["x"], g1["steps_int"], tr["noise"], context=[g1["ctx"]])
    assert rel_err(pred, tr["pred"]) < TOL and abs(float(loss) - float(tr["loss"])) < 1e-6
    assert set(grads) == set(tr["grads"])
    for k, g in tr["grads"].items():
        assert rel_err(grads[k], g) < 2e-5, k
    tr = g2["train"]
    pred, loss, grads = ref.train_loss_and_grads(g2["state_dict"], g2["x"], g2["steps"], tr["noise"], input_data=g2["input_data"])
    assert rel_err(pred, tr["pred"]) < TOL and abs(float(loss) - float(tr["loss"])) < 1e-6
    assert set(grads) == set(tr["grads"])
    for k, g in tr["grads"].items():
        assert rel_err(grads[k], g) < 2e-5, k


def test_ddim_table_and_timesteps():
    """Known values of the restated schedule (parity unpinned vs diffusers)."""
    acp = ddim_ref.alphas_cumprod()
    assert acp.dtype == torch.float32 and acp.shape == (1000,)
    for i, v in ((0, 0.99995869), (20, 0.99811417), (500, 0.49228504), (980, 8.7653e-04)):
        assert abs(float(acp[i]) - v) / v < 2e-5
    assert ddim_ref.timesteps(50).tolist() == list(range(980, -1, -20))
    assert ddim_ref.timesteps(30).tolist()[0] == 957 and ddim_ref.timesteps(30).tolist()[-1] == 0
    assert ddim_ref.timesteps(10).tolist() == list(range(900, -1, -100))


def test_ddim_step_inverts_add_noise():
    """Property: with the true eps, one step from t lands on add_noise(x0, eps, t_prev)."""
    acp = ddim_ref.alphas_cumprod()
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(3, 16, 20, generator=g, dtype=torch.float64)
    eps = torch.randn(3, 16, 20, generator=g, dtype=torch.float64)
    for t in (980, 500, 20):
        xt = ddim_ref.add_noise(x0, eps, torch.full((3,), t), acp.double())
        prev = ddim_ref.step(eps, t, xt, 50, acp.double())
        want = ddim_ref.add_noise(x0, eps, torch.full((3,), t - 20), acp.double())
        assert rel_err(prev, want) < 1e-9
    x_last = ddim_ref.step(eps, 0, ddim_ref.add_noise(x0, eps, torch.zeros(3, dtype=torch.long), acp.double()), 50, acp.double())
    assert rel_err(x_last, x0) < 1e-9  # final_alpha_cumprod = 1
