"""The forms of the fused-run kernel against each other, bit for bit.

chain4.hip carries hand-scheduled inline-asm k-loops and whole slots (chain4_kloop.inc, chain4_fused.inc: generated) whose correctness rests
on hand-counted waits and on registers the compiler does not know about; its two-role slots and its generic row phases are plain HIP C++
(pinned against the reference goldens by tests/test_hip_model.py).  Every arithmetic step is the same (MFMA order per accumulator,
power-of-two row scales, split), so every saved activation, row maximum, weight / bias / input gradient and inference output must be
IDENTICAL between
  (default)            hot slots as fused statements
  PAPR_C4_FUSED=0      two-role slots everywhere (one-statement k-loops, C++ row phases with the flags as constants)
  PAPR_C4_GENERIC=1    ... and the generic row phases (flags looked at at run time)
  PAPR_C4_PAIRS=0      every hot slot its own statement instead of two slots per statement (forward runs)
  PAPR_C4_DMA=1        the run's input rows split ahead of it (round 4: split_rows_kernel) and staged by LDS-DMA, not by the run itself
and from run to run (the races this file guards against show up as run-to-run differences).  Sizes: a cloud-sized M with ragged last
tiles and several workgroup iterations, and one below a tile; ReLU and LeakyReLU."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (PAPR_H3_ROWS=f32: the parity mode with fp32 rows between a run and its weight gradients, PAPR_MLP_H3 -- the default since round 6 keeps f16 rows, whose
#  forms are compared by test_f16_rows_mode_is_h3_but_for_the_weight_gradients below)
F32 = {"PAPR_H3_ROWS": "f32"}
VARIANTS = [("generic rows", dict(F32, PAPR_C4_GENERIC="1")), ("two-role", dict(F32, PAPR_C4_FUSED="0")), ("fused", dict(F32)), ("fused again", dict(F32)),
            # every hot slot its own statement (the default runs tile Y's step and tile X's next step of a forward run as ONE statement)
            ("fused, single slots", dict(F32, PAPR_C4_PAIRS="0")),
            # the run takes its input rows split ahead of it (split_rows_kernel) through LDS-DMA instead of splitting them itself while it stages
            # them (registers + vector instructions): the same arithmetic, instruction for instruction
            ("rows split ahead", dict(F32, PAPR_C4_DMA="1")), ("rows split ahead, generic rows", dict(F32, PAPR_C4_DMA="1", PAPR_C4_GENERIC="1"))]


def _run(tmp_path, name, env, M, n, act, dims=()):
    out = tmp_path / (name.replace(" ", "_") + ".pt")
    e = {k: v for k, v in os.environ.items() if k not in ("PAPR_C4_GENERIC", "PAPR_C4_FUSED", "PAPR_GEMM_MODE", "PAPR_C4_DMA", "PAPR_C4_PAIRS", "PAPR_H3_ROWS", "PAPR_TN_TR", "PAPR_VARIANT_TOP_F16", "PAPR_VARIANT_GAIN", "PAPR_VARIANT_GRAD_SCALE", "PAPR_VARIANT_SKIP")}
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "chain_variants_worker.py"), str(out), str(M), str(n), act] + [str(v) for v in dims], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (name, r.stderr[-2000:])
    return torch.load(out)


H1_VARIANTS = [("h1 generic rows", {"PAPR_C4_GENERIC": "1", "PAPR_GEMM_MODE": "h1"}), ("h1", {"PAPR_GEMM_MODE": "h1"}), ("h1 again", {"PAPR_GEMM_MODE": "h1"}),
               ("h1 single slots", {"PAPR_GEMM_MODE": "h1", "PAPR_C4_PAIRS": "0"}),
               # the top gradient rows arrive as papr_f16_rows (as papr_attn_tail_bwd writes them) and are staged by LDS-DMA
               ("h1 top rows f16", {"PAPR_GEMM_MODE": "h1", "PAPR_VARIANT_TOP_F16": "1"}),
               ("h1 top rows f16, generic rows", {"PAPR_GEMM_MODE": "h1", "PAPR_VARIANT_TOP_F16": "1", "PAPR_C4_GENERIC": "1"})]


def _flat(res):
    for k, v in sorted(res.items()):
        if isinstance(v, list):
            for i, t in enumerate(v):
                yield "%s[%d]" % (k, i), t
        else:
            yield k, v


# (the last case: the value MLP's shape -- 141 input columns, 32 output columns -- and a last tile of 37 rows)
@pytest.mark.parametrize("M,n,act,dims", [(40000, 5, "relu", ()), (45, 3, "leakyrelu", ()), (20000, 4, "leakyrelu", ()), (30053, 4, "relu", (141, 32))])
def test_fused_run_kernels_agree_bit_for_bit(tmp_path, M, n, act, dims):
    ref = None
    for name, env in VARIANTS:
        res = _run(tmp_path, name, env, M, n, act, dims)
        if ref is None:
            ref = res
            continue
        for (k, a), (_, b) in zip(_flat(ref), _flat(res)):
            assert a.shape == b.shape, (name, k)
            same = torch.equal(a, b)
            assert same, "%s: %s differs from the generic form in %d of %d elements (max |diff| %g)" % (name, k, int((a != b).sum()), a.numel(), float((a - b).abs().max()))


@pytest.mark.parametrize("M,n,act,dims", [(40000, 5, "relu", ()), (30053, 4, "leakyrelu", (141, 32)), (45, 3, "leakyrelu", ())])
def test_one_product_mode_forms_agree_bit_for_bit(tmp_path, M, n, act, dims):
    """The same for the reduced-precision mode (PAPR_GEMM_MODE=h1: two-role slots; hot and generic row phases)."""
    ref = None
    for name, env in H1_VARIANTS:
        res = _run(tmp_path, name, env, M, n, act, dims)
        if ref is None:
            ref = res
            continue
        for (k, a), (_, b) in zip(_flat(ref), _flat(res)):
            assert torch.equal(a, b), "%s: %s differs in %d of %d elements" % (name, k, int((a != b).sum()), a.numel())


F16ROWS_VARIANTS = [("h3_f16rows generic rows", {"PAPR_C4_GENERIC": "1", "PAPR_GEMM_MODE": "h3_f16rows"}), ("h3_f16rows", {}),      # ({}: the default IS this mode)
                    ("h3_f16rows again", {"PAPR_GEMM_MODE": "h3_f16rows"}), ("h3_f16rows single slots", {"PAPR_GEMM_MODE": "h3_f16rows", "PAPR_C4_PAIRS": "0"}),
                    ("h3_f16rows two-role", {"PAPR_GEMM_MODE": "h3_f16rows", "PAPR_C4_FUSED": "0"}),
                    # the top gradient rows arrive split (papr_f16_rows with a lo plane, as papr_attn_tail_bwd writes them) and are staged by LDS-DMA
                    ("h3_f16rows top rows split", {"PAPR_VARIANT_TOP_F16": "1"}), ("h3_f16rows top rows split, generic rows", {"PAPR_VARIANT_TOP_F16": "1", "PAPR_C4_GENERIC": "1"})]


@pytest.mark.parametrize("M,n,act,dims", [(40000, 5, "relu", ()), (30053, 4, "leakyrelu", (141, 32)), (45, 3, "leakyrelu", ()), (45, 3, "relu", (141, 32))])       # (the last two: less than a tile, less than a stage of the weight-gradient kernels)
def test_f16_rows_mode_is_h3_but_for_the_weight_gradients(tmp_path, M, n, act, dims):
    """PAPR_MLP_H3_F16ROWS (round 6's gated experiment): the parity arithmetic whose fused runs keep f16 rows for their weight gradients.  Its forms agree
    bit for bit; against the default mode the run's result, the inference pass and the input gradient are IDENTICAL (the forward and data-gradient
    arithmetic is untouched), weight and bias gradients carry the f16 rounding of their operands' rows: the measured distance is printed, the bar is the
    one the default mode's gradients are held to against the reference (rms within 1.5e-4 of the tensor's largest element, tests/conftest.py)."""
    base = _run(tmp_path, "h3", dict(F32), M, n, act, dims)
    ref = None
    for name, env in F16ROWS_VARIANTS:
        res = _run(tmp_path, name, env, M, n, act, dims)
        if ref is None:
            ref = res
            continue
        for (k, a), (_, b) in zip(_flat(ref), _flat(res)):
            assert a.shape == b.shape, (name, k)
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), "%s: %s differs in %d of %d elements" % (name, k, int((a.view(torch.int32) != b.view(torch.int32)).sum()), a.numel())
    # the weight gradients of the full 256 x 256 layers run on the LDS-DMA + transposing-read kernel (gemm_tn_tr_kernel; round 6, the default since its
    # second version) -- against the register-staged kernel (PAPR_TN_TR=0): the same rows, slices and matrix instructions; ONE operand carries both rows'
    # scales instead of each its own, so the f16 roundings of rows far below the slice's largest differ -- equal to 1e-5 of the tensor's maximum, not bit for bit
    tr = _run(tmp_path, "h3_f16rows register-staged weight gradients", {"PAPR_TN_TR": "0"}, M, n, act, dims)
    for k in ("d_ws", "d_bs"):
        for i, (a, b) in enumerate(zip(ref[k], tr[k])):
            rel = float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))
            assert rel < 1e-5, (k, i, rel)
    assert torch.equal(ref["d_x"], tr["d_x"])
    for k in ("d_x", "inf"):                      # (d_x2 / d_ws2 -- a backward pass WITHOUT the forward pass's saved state -- read the inner rows as fp32 masks: not a
                                                  #  path of this mode or of h1, whose inner rows are f16; compared among the mode's own forms above only)
        assert torch.equal(base[k], ref[k]), "%s differs from the default mode's in %d elements" % (k, int((base[k] != ref[k]).sum()))
    assert torch.equal(base["outs"][-1], ref["outs"][-1])
    worst = 0.0
    for k in ("d_ws", "d_bs"):
        for i, (a, b) in enumerate(zip(base[k], ref[k])):
            rel = float((a - b).pow(2).mean().sqrt() / a.abs().max().clamp_min(1e-30))
            worst = max(worst, rel)
            print("%s[%d]: rms |f16rows - h3| / max |h3| = %.3g" % (k, i, rel))
            assert rel < 1.5e-4, (k, i, rel)
    print("worst %.3g" % worst)


def test_one_product_weight_gradients_when_inner_rows_outgrow_the_run_scale(tmp_path):
    """The one-product mode keeps ONE scale per row and run: a layer's rows inside the run may be up to 2^12 larger (2^9 when this was found) than the rows the scale was taken from.
    Weights three times the initialisation's (activations and gradients grow ~20-fold through the run): the weight gradients stay finite and within the
    mode's tolerance of the parity arithmetic's, on both weight-gradient kernels.  (Round 6: the register-staged kernel took its slice scale from the top
    rows' maxima and overflowed f16 once a trained network's inner gradients were eight times its top gradients -- inf in dW on every step, the GradScaler
    at 2^-50 from step ~5,000 of a chair.yml run under use_amp.)"""
    M, n, act = 20000, 5, "relu"
    ref = _run(tmp_path, "parity", dict(F32, PAPR_VARIANT_GAIN="3"), M, n, act)
    for name, env in (("h1 gain", {"PAPR_GEMM_MODE": "h1", "PAPR_VARIANT_GAIN": "3"}), ("h1 gain register-staged", {"PAPR_GEMM_MODE": "h1", "PAPR_VARIANT_GAIN": "3", "PAPR_TN_TR": "0"})):
        res = _run(tmp_path, name, env, M, n, act)
        for k in ("d_ws", "d_bs", "d_x"):
            for i, (a, b) in enumerate(zip(ref[k] if isinstance(ref[k], list) else [ref[k]], res[k] if isinstance(res[k], list) else [res[k]])):
                assert torch.isfinite(b).all(), (name, k, i, "not finite")
                rel = float((a - b).double().pow(2).mean().sqrt() / a.abs().max().clamp_min(1e-30))        # (the mode's own bar on a render's gradients is 1e-2: tests/test_hip_h1.py)
                print(name, k, i, "rms error / max |reference| = %.2e" % rel)
                assert rel < 3e-2, (name, k, i, rel)      # (measured 8e-5 ... 1.0e-2 on these rows -- 24 powers of two apart, zero rows among them; with the initialisation's weights the same)


def test_one_product_data_gradient_run_hands_inf_and_nan_on(tmp_path):
    """The GradScaler's overflow signal: top gradient rows that hold inf or nan must leave a one-product data-gradient run as non-finite gradients (round 6 tried
    MODE.FP16_OVFL to make rows beyond the run scale's headroom saturate -- it also turned a row of nans into finite numbers; dropped)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "inf_rows_worker.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = {l.split(" | ")[0]: l for l in r.stdout.splitlines() if " | " in l}
    for label in ("all inf", "one inf per row", "all nan"):
        assert "d_x finite elements: 0 of" in lines[label], lines[label]


# (the parity arithmetic only: a one-product run clamps its gradient rows' scale at 2^-40 -- rows this small vanish there by design, a GradScaler keeps them away)
@pytest.mark.parametrize("mode", ["h3_f16rows"])
def test_weight_gradients_of_tiny_gradient_rows_stay_finite(tmp_path, mode):
    """Gradient rows of ~1e-33 and below (the query MLP late in a run with few points): the transposing weight-gradient kernel's factor (1 / scale_g)(1 / scale_x)
    leaves fp32's normal range as a product -- it is kept as a sum of exponents.  (Round 6: the product underflowed, its reciprocal was inf, the weight gradients
    NaN: a lego.yml run's parameters at step 13,904.)  Against the register-staged kernel: finite, and equal to 1e-5 of the largest element."""
    M, n, act = 20000, 5, "relu"
    env = {"PAPR_GEMM_MODE": mode, "PAPR_VARIANT_GRAD_SCALE": "1e-33"}
    tr = _run(tmp_path, mode + " tiny", env, M, n, act)
    reg = _run(tmp_path, mode + " tiny register-staged", dict(env, PAPR_TN_TR="0"), M, n, act)
    for k in ("d_ws", "d_bs"):
        for i, (a, b) in enumerate(zip(reg[k], tr[k])):
            assert torch.isfinite(b).all() and torch.isfinite(a).all(), (k, i)
            assert float(a.abs().max()) > 0, (k, i, "the reference itself vanished")
            rel = float((a - b).abs().max() / a.abs().max())
            assert rel < 1e-5, (k, i, rel)
