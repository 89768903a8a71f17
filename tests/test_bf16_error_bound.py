"""north_star's tolerance for the bf16 configs, stated against the fp32 REFERENCE formula and measured at full size
(r2 only compared the bf16 kernels with the oracle's bf16-quantised mode and with a 2e-3 bound).

BASELINE configs[2] (480p, T=5, 4 ids) and configs[4] (720p, T=10, 6 ids), whole frame, fp32 embeddings handed to
every mode.  Reference = the fp32 MFMA kernel, which is bit-exact against the pinned oracle (test_gpu_global.py) and is
spot-checked against the oracle again here on the same tensors.  Figures are on the normalised maps
(sigmoid(d) - 0.5) * 2 -- what the segmentation head consumes (IntVOS.py:611-612).

Measured on MI355X (tools/bf16_error.py, r3):       scale 0.1 (bench.py's distribution)    scale 0.3
    compute="bf16"    cfg3  max |err|                    7.0e-4                           1.4e-3
                      cfg5                               9.1e-4                           1.7e-3
    compute="bf16x3"  cfg3 / cfg5                        2.3e-6 / 2.4e-6                  4.8e-6 / 6.4e-6
    compute="bf16r"   (bf16 filter + exact fp32 re-rank) 0 (bit-equal to the fp32 kernel) 0
So: plain bf16 meets the 1e-3 bar on the benchmark's embedding distribution but NOT in general (its error scales with
|q||k|: inputs are rounded to 8 significand bits); the modes that meet 1e-3 whatever the inputs are bf16x3 and bf16r.
The bounds asserted below are those statements, with the measured headroom written next to each."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = {3: (120, 214, 5, 4), 5: (180, 320, 10, 6)}
TOL = 1e-3  # north_star: "within 1e-3 of the reference"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cvpr2020_manet_amd import ops as o
    return o


def _inputs(cfg, scale):
    H, W, T, n_ids = CFG[cfg]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(20200614 + cfg)
    cur = torch.relu(torch.randn(100, H, W, generator=g, device=dev)) * scale
    bank = torch.relu(torch.randn(T * H * W, 100, generator=g, device=dev)) * scale
    lab = torch.randint(0, n_ids, (T * H * W,), generator=g, device=dev, dtype=torch.int32)
    return cur.permute(1, 2, 0), bank, lab, n_ids


def _norm(x):
    return (torch.sigmoid(x) - 0.5) * 2


@pytest.mark.parametrize("cfg", [3, 5])
@pytest.mark.parametrize("scale", [0.1, 0.3])
def test_error_vs_fp32_reference_at_full_size(ops, oracle, cfg, scale):
    q, bank, lab, n_ids = _inputs(cfg, scale)
    ref = ops.global_match(bank, q, lab, n_ids, compute="f32")
    # the fp32 kernel IS the reference formula: spot-check it against the pinned oracle on these very tensors
    nq = 256
    qs = q.reshape(-1, 100)[:nq].cpu().numpy().reshape(nq, 1, 100)
    want = oracle.global_match(bank.cpu().numpy().reshape(-1, 1, 100), qs, lab.cpu().numpy().reshape(-1, 1, 1), 1,
                               n_ids=n_ids).reshape(nq, n_ids)
    assert np.array_equal(ref[:nq].cpu().numpy(), want)
    refn = _norm(ref)
    errs = {}
    for mode in [m for m in ("bf16", "bf16x3", "bf16r") if m in ops.COMPUTE]:
        got = ops.global_match(bank, q, lab, n_ids, compute=mode)
        errs[mode] = ((_norm(got) - refn).abs().max().item(), (got - ref).abs().max().item(),
                      (got.argmin(1) != ref.argmin(1)).float().mean().item())
    print("cfg%d scale %.1f: (normalised max err, raw max err, arg-min id flips) %s" % (cfg, scale, errs))
    # split-bf16: fp32-class, everywhere (measured <= 6.5e-6: 150x headroom)
    assert errs["bf16x3"][0] <= TOL / 100
    if "bf16r" in errs:  # bf16 filter + exact fp32 re-rank: bit-equal to the fp32 kernel
        assert errs["bf16r"][1] == 0.0 and errs["bf16r"][2] == 0.0
    if scale == 0.1:
        # plain bf16 on the benchmark's distribution: inside the bar (measured 7.0e-4 / 9.1e-4)
        assert errs["bf16"][0] <= TOL
    else:
        # ... and outside it on 3x larger embeddings (measured 1.4e-3 / 1.7e-3): documented, bounded, not hidden
        assert TOL < errs["bf16"][0] <= 2.5e-3
    # rounding the inputs moves a distance by a few 1e-3 relative; the arg-min object changes only where two objects'
    # distances are that close (random labels: < 1 % of the pixels)
    assert errs["bf16"][2] < 0.02 and errs["bf16x3"][2] < 1e-3
