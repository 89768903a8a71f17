"""gradients of the matching path against what torch.autograd gives the reference.
`tests/golden/grad_tiny.npz` was produced by oracle/gen_golden.py:variant_grad, which imports the reference and
calls torch.autograd.grad / loss.backward() on its functions and on a whole IntVOS.forward training step
(train_stage1.py:126-156 shape) with tiny heads.  Tolerances: fp32 rounding through ~10 ops (rtol 2e-4) -- the
reference's own torch.matmul / pow / sum orders differ from the kernels' fmaf chains."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from test_intvos_module import TinyExtractor

pytestmark = pytest.mark.gpu
RTOL, ATOL = 2e-4, 2e-6


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cvpr2020_manet_amd import ops as o
    return o


@pytest.fixture(scope="module")
def g():
    return load_golden("grad_tiny")


def dev(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.requires_grad_(True) if grad else t


def test_global_match_gradients_match_reference_autograd(ops, g):
    ref, qry = dev(g["g_ref_chw"], True), dev(g["g_qry_chw"], True)
    lab = dev(g["g_labels"])
    out = ops.global_match(ref.permute(1, 2, 0), qry.permute(1, 2, 0), lab, 4, normalize=True)
    assert out.requires_grad
    # forward: same values as the no-grad kernel, bit for bit
    with torch.no_grad():
        plain = ops.global_match(ref.permute(1, 2, 0), qry.permute(1, 2, 0), lab, 4, normalize=True)
    assert torch.equal(out.detach(), plain)
    w = dev(g["g_weight"]).reshape(out.shape)
    gr, gq = torch.autograd.grad((out * w).sum(), [ref, qry])
    np.testing.assert_allclose(gr.cpu().numpy(), g["g_grad_ref"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(gq.cpu().numpy(), g["g_grad_qry"], rtol=RTOL, atol=ATOL)
    # raw distances, 3 ids
    out = ops.global_match(ref.permute(1, 2, 0), qry.permute(1, 2, 0), lab, 3)
    w = dev(g["g_weight_raw"]).reshape(out.shape)
    gr, gq = torch.autograd.grad((out * w).sum(), [ref, qry])
    np.testing.assert_allclose(gr.cpu().numpy(), g["g_grad_ref_raw"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(gq.cpu().numpy(), g["g_grad_qry_raw"], rtol=RTOL, atol=ATOL)


def test_global_match_arg_is_the_first_nearest_row(ops):
    """the recorded bank row really attains the minimum; ties resolve to the first row of the object"""
    from cvpr2020_manet_amd.autograd import GlobalMatchFn
    torch.manual_seed(3)
    C, N, M = 100, 700, 900
    q = torch.relu(torch.randn(N, C, device="cuda")) * 0.1
    k = torch.relu(torch.randn(M, C, device="cuda")) * 0.1
    k[500:520] = k[100:120]  # duplicated rows: exact ties
    lab = torch.randint(-1, 3, (M,), device="cuda", dtype=torch.int32)
    lab[500:520] = lab[100:120]
    out, arg = GlobalMatchFn.apply(k, q, lab, 4)
    d = (q.double() ** 2).sum(1, keepdim=True) + (k.double() ** 2).sum(1)[None] - 2 * q.double() @ k.double().t()
    for o in range(4):
        rows = arg[:, o].long()
        if not (lab == o).any():
            assert torch.all(rows == -1) and torch.all(out[:, o] == 1e20)
            continue
        assert torch.all(lab[rows] == o)
        dm = d.masked_fill((lab != o)[None], float("inf"))
        assert torch.allclose(dm.gather(1, rows[:, None])[:, 0], dm.min(1).values, rtol=0, atol=2e-6)
        assert not ((rows >= 500) & (rows < 520)).any()  # the earlier twin wins a tie


@pytest.mark.parametrize("i", [0, 1, 2])
def test_local_match_gradients_match_reference_autograd(ops, g, i):
    prev, cur = dev(g["l%d_prev_chw" % i], True), dev(g["l%d_cur_chw" % i], True)
    lab, d, n_ids = dev(g["l%d_labels" % i]), int(g["l%d_d" % i]), int(g["l%d_n_ids" % i])
    out = ops.local_match(prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab, n_ids, d)
    assert out.requires_grad
    np.testing.assert_allclose(out.detach().cpu().numpy().reshape(g["l%d_out" % i].shape), g["l%d_out" % i],
                               rtol=1e-5, atol=2e-6)
    with torch.no_grad():  # the training forward equals the fused inference kernel bit for bit
        assert torch.equal(out.detach(), ops.local_match(prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab, n_ids, d))
    w = dev(g["l%d_weight" % i]).reshape(out.shape)
    gp, gc = torch.autograd.grad((out * w).sum(), [prev, cur])
    scale = max(np.abs(g["l%d_grad_cur" % i]).max(), 1e-6)
    np.testing.assert_allclose(gp.cpu().numpy(), g["l%d_grad_prev" % i], rtol=RTOL, atol=2e-5 * scale)
    np.testing.assert_allclose(gc.cpu().numpy(), g["l%d_grad_cur" % i], rtol=RTOL, atol=2e-5 * scale)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_local_match_no_downsample_gradients_match_reference_autograd(ops, i):
    """MODEL_LOCAL_DOWNSAMPLE False in training (IntVOS.py:299-313 raw full-resolution distances; :398-432
    stride-2 label gather, constant 1.0): forward values and both gradients of the reference's own autograd
    (tests/golden/grad_ds0.npz, oracle/gen_golden.py:variant_grad_ds0)"""
    g0 = load_golden("grad_ds0")
    prev, cur = dev(g0["l%d_prev_chw" % i], True), dev(g0["l%d_cur_chw" % i], True)
    lab, d, n_ids = dev(g0["l%d_labels" % i]), int(g0["l%d_d" % i]), int(g0["l%d_n_ids" % i])
    out = ops.local_match(prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab, n_ids, d, downsample=False)
    assert out.requires_grad
    np.testing.assert_allclose(out.detach().cpu().numpy().reshape(g0["l%d_out" % i].shape), g0["l%d_out" % i], rtol=1e-5, atol=2e-6)
    with torch.no_grad():  # the training forward equals the inference path
        assert torch.equal(out.detach(), ops.local_match(prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab, n_ids, d, downsample=False))
    w = dev(g0["l%d_weight" % i]).reshape(out.shape)
    gp, gc = torch.autograd.grad((out * w).sum(), [prev, cur])
    scale = max(np.abs(g0["l%d_grad_cur" % i]).max(), 1e-6)
    np.testing.assert_allclose(gp.cpu().numpy(), g0["l%d_grad_prev" % i], rtol=RTOL, atol=2e-5 * scale)
    np.testing.assert_allclose(gc.cpu().numpy(), g0["l%d_grad_cur" % i], rtol=RTOL, atol=2e-5 * scale)


def test_training_step_through_forward_matches_reference(ops, g):
    """IntVOS.forward in train() mode + backward: logits and parameter gradients of the reference's own run"""
    from cvpr2020_manet_amd.config import make_cfg
    from cvpr2020_manet_amd.networks import IntVOS as M
    cfg = make_cfg(["--TEST_MODE", "False", "--MODEL_SEMANTIC_EMBEDDING_DIM", "12", "--MODEL_HEAD_EMBEDDING_DIM", "8",
                    "--MODEL_ASPP_OUTDIM", "6", "--MODEL_MAX_LOCAL_DISTANCE", "2"])
    model = M.IntVOS(cfg, TinyExtractor())
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")}
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    nobj = int(g["t_nobj"])
    dic = model.forward(dev(g["t_x"]), dev(g["t_ref_lab"]), dev(g["t_prev_lab"]), seq_names=["clip"],
                        gt_ids=torch.Tensor([nobj]), k_nearest_neighbors=1, global_map_tmp_dic=None,
                        local_map_dics=None, interaction_num=1, start_annotated_frame=0, frame_num=[2])
    logits = dic["clip"]
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["t_logits"], rtol=1e-3, atol=1e-4)
    (logits * dev(g["t_wl"])).sum().backward()
    params = dict(model.named_parameters())
    for name in g["t_grad_names"].tolist():
        want = g["t_grad::" + name]
        got = params[name].grad.cpu().numpy()
        assert np.abs(got).max() > 0  # the gradient really arrived (r1: silently dropped)
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4 * max(np.abs(want).max(), 1e-6))


def test_training_step_outside_the_defaults_matches_reference(ops):
    """the same step with k_nearest_neighbors = 3 and MODEL_LOCAL_DOWNSAMPLE False (r5: both raised before): logits and parameter
    gradients of the reference's own run (tests/golden/grad_step_alt.npz)"""
    from cvpr2020_manet_amd.config import make_cfg
    from cvpr2020_manet_amd.networks import IntVOS as M
    g = load_golden("grad_step_alt")
    cfg = make_cfg(["--TEST_MODE", "False", "--MODEL_SEMANTIC_EMBEDDING_DIM", "12", "--MODEL_HEAD_EMBEDDING_DIM", "8",
                    "--MODEL_ASPP_OUTDIM", "6", "--MODEL_MAX_LOCAL_DISTANCE", "2", "--MODEL_LOCAL_DOWNSAMPLE", "False"])
    model = M.IntVOS(cfg, TinyExtractor())
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")}
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    nobj, knn = int(g["t_nobj"]), int(g["t_knn"])
    dic = model.forward(dev(g["t_x"]), dev(g["t_ref_lab"]), dev(g["t_prev_lab"]), seq_names=["clip"],
                        gt_ids=torch.Tensor([nobj]), k_nearest_neighbors=knn, global_map_tmp_dic=None,
                        local_map_dics=None, interaction_num=1, start_annotated_frame=0, frame_num=[2])
    logits = dic["clip"]
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["t_logits"], rtol=1e-3, atol=1e-4)
    (logits * dev(g["t_wl"])).sum().backward()
    params = dict(model.named_parameters())
    for name in g["t_grad_names"].tolist():
        want = g["t_grad::" + name]
        got = params[name].grad.cpu().numpy()
        assert np.abs(got).max() > 0
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4 * max(np.abs(want).max(), 1e-6))


def _corr_reference(a, b, pad, K, md, s1, s2):
    """differentiable torch restatement of the correlation_package forward contract (correlation_cuda_kernel.cu:73-147)"""
    B, C, H, W = a.shape
    kr, r = (K - 1) // 2, md // s2
    ap = nn.functional.pad(a, (pad,) * 4)
    bp = nn.functional.pad(b, (pad,) * 4)
    ph, pw = H + 2 * pad, W + 2 * pad
    border = kr + md
    oh = -(-(ph - 2 * border) // s1)
    ow = -(-(pw - 2 * border) // s1)
    outs = []
    for tj in range(-r, r + 1):
        for ti in range(-r, r + 1):
            acc = 0
            for j in range(-kr, kr + 1):
                for i in range(-kr, kr + 1):
                    ys = torch.arange(oh) * s1 + md + j
                    xs = torch.arange(ow) * s1 + md + i
                    pa = ap[:, :, ys][:, :, :, xs]
                    pb = bp[:, :, ys + tj * s2][:, :, :, xs + ti * s2]
                    acc = acc + (pa * pb).sum(1)
            outs.append(acc / (K * K * C))
    return torch.stack(outs, 1)


@pytest.mark.parametrize("cfgc", [(2, 5, 9, 8, 3, 3, 2, 2, 2), (1, 12, 10, 11, 2, 1, 2, 1, 1), (1, 7, 12, 12, 6, 1, 6, 1, 2)])
def test_correlation_backward_vs_torch_autograd(ops, cfgc):
    B, C, H, W, pad, K, md, s1, s2 = cfgc
    torch.manual_seed(11)
    a = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    b = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    out = ops.correlation_forward(a, b, pad, K, md, s1, s2)
    ref = _corr_reference(a, b, pad, K, md, s1, s2)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    w = torch.randn_like(out)
    ga, gb = torch.autograd.grad((out * w).sum(), [a, b])
    ra, rb = torch.autograd.grad((ref * w).sum(), [a, b])
    torch.testing.assert_close(ga, ra, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(gb, rb, rtol=1e-4, atol=1e-6)


def test_global_match_knn_gradients_match_reference_autograd(ops):
    """k_nearest_neighbors > 1 in training (IntVOS.py:87-94).  tests/golden/grad_knn.npz = the reference's
    own autograd through topk -> where(valid, d, farthest real neighbour) -> mean (oracle/gen_golden.py:variant_grad_knn), on
    banks with an object of fewer than k pixels (the padding rule carries gradient to the farthest real one), an object with
    one pixel and one with none (no gradient)."""
    g = load_golden("grad_knn")
    for i in range(int(g["n_cases"])):
        ref, qry = dev(g["c%d_ref_chw" % i], True), dev(g["c%d_qry_chw" % i], True)
        lab, k, n_ids = dev(g["c%d_labels" % i]), int(g["c%d_k" % i]), int(g["c%d_n_obj" % i]) + 1
        out = ops.global_match(ref.permute(1, 2, 0), qry.permute(1, 2, 0), lab, n_ids, k_nearest_neighbors=k)
        assert out.requires_grad
        want = g["c%d_out" % i].reshape(-1, n_ids)
        np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=2e-5, atol=2e-6)
        with torch.no_grad():  # ... and the inference kernel's top-k gives the same values
            plain = ops.global_match(ref.permute(1, 2, 0), qry.permute(1, 2, 0), lab, n_ids, k_nearest_neighbors=k)
        torch.testing.assert_close(out.detach(), plain, rtol=1e-6, atol=1e-6)
        w = dev(g["c%d_weight" % i]).reshape(out.shape)
        norm = (torch.sigmoid(out) - 0.5) * 2
        gr, gq = torch.autograd.grad((norm * w).sum(), [ref, qry])
        np.testing.assert_allclose(gr.cpu().numpy(), g["c%d_grad_ref" % i], rtol=RTOL, atol=ATOL, err_msg="case %d ref" % i)
        np.testing.assert_allclose(gq.cpu().numpy(), g["c%d_grad_qry" % i], rtol=RTOL, atol=ATOL, err_msg="case %d qry" % i)


def test_topk_arg_passes_are_exact_and_ordered(ops):
    """manet_global_match_topk_arg_f32 against a brute-force top-k: the k smallest distances per (query, object) in ascending
    order with the rows that attain them; past an object's row count 1e20 / -1; exact duplicates of a row appear once each"""
    import ctypes
    from cvpr2020_manet_amd import _lib
    lib = _lib.load()
    torch.manual_seed(5)
    C, N, M, n_ids, k = 100, 300, 500, 4, 6
    q = torch.relu(torch.randn(N, C, device="cuda")) * 0.3
    b = torch.relu(torch.randn(M, C, device="cuda")) * 0.3
    b[17] = b[3]  # an exact duplicate row (same label below): both must be listed, once each
    lab = torch.randint(0, 2, (M,), device="cuda", dtype=torch.int32)
    lab[17] = lab[3]
    lab[100:103] = 2  # object 2: three rows (< k); object 3: none
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.manet_global_match_topk_arg_workspace_bytes(N, M, C, n_ids, ctypes.byref(nbytes)), "ws")
    ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device="cuda")
    d = torch.empty(k, N, n_ids, device="cuda")
    arg = torch.empty(k, N, n_ids, dtype=torch.int32, device="cuda")
    rc = lib.manet_global_match_topk_arg_f32(q.data_ptr(), C, 1, b.data_ptr(), C, 1, lab.data_ptr(), N, M, C, n_ids, k,
                                             d.data_ptr(), arg.data_ptr(), ws.data_ptr(), ws.numel(),
                                             torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "manet_global_match_topk_arg_f32")
    full = ((q * q).sum(1, keepdim=True) + (b * b).sum(1)[None] - 2 * q @ b.t()).double()
    for o in range(n_ids):
        rows = torch.nonzero(lab == o).flatten()
        v = min(k, rows.numel())
        if v:
            top, idx = torch.topk(full[:, rows], v, dim=1, largest=False)
            got_d = d[:v, :, o].t().double()
            torch.testing.assert_close(got_d, top, rtol=1e-5, atol=2e-6)
            got_rows = arg[:v, :, o].t().long()
            assert bool((lab[got_rows.flatten()] == o).all())
            # the listed rows attain the listed distances, and no row is listed twice for a query
            torch.testing.assert_close(torch.gather(full, 1, got_rows), got_d, rtol=1e-5, atol=2e-6)
            srt = torch.sort(got_rows, dim=1).values
            assert v == 1 or bool((srt[:, 1:] != srt[:, :-1]).all())
            assert bool((d[1:v, :, o] >= d[:v - 1, :, o]).all())
        assert bool((arg[v:, :, o] == -1).all()) and bool((d[v:, :, o] >= 1e20).all())


def test_unsupported_training_configurations_raise(ops):
    q = (torch.rand(6, 7, 16, device="cuda")).requires_grad_(True)
    k = torch.rand(6, 7, 16, device="cuda")
    lab = torch.zeros(6, 7, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="backward exists"):
        ops.global_match(k, q, lab, 2, compute="bf16")
    with pytest.raises(RuntimeError, match="backward exists"):
        ops.global_match(k, q, lab, 2, k_nearest_neighbors=9)


def test_correlation_classes_keep_the_reference_interface_and_dtypes(ops):
    """correlation_package/correlation.py:7-61: Correlation(pad, K, max_disp, s1, s2, corr_multiply)(in1, in2) and
    CorrelationFunction(...)(in1, in2); output (and gradients) in the inputs' dtype -- float, half and double, forward and
    backward (VERDICT r2 "next" #7, ADVICE r2: the autograd path used to widen half inputs to float; r4: double backward)."""
    from cvpr2020_manet_amd.correlation import Correlation, CorrelationFunction
    torch.manual_seed(2)
    a = torch.randn(2, 6, 9, 11, device="cuda")
    b = torch.randn(2, 6, 9, 11, device="cuda")
    mod = Correlation(pad_size=3, kernel_size=1, max_displacement=3, stride1=1, stride2=1, corr_multiply=1)
    assert (mod.pad_size, mod.kernel_size, mod.max_displacement, mod.stride1, mod.stride2, mod.corr_multiply) == (3, 1, 3, 1, 1, 1)
    with torch.no_grad():
        want = _corr_reference(a, b, 3, 1, 3, 1, 1)
        torch.testing.assert_close(mod(a, b), want, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(CorrelationFunction(3, 1, 3, 1, 1, 1)(a, b), want, rtol=1e-5, atol=1e-6)
        assert mod(a.half(), b.half()).dtype == torch.float16 and mod(a.double(), b.double()).dtype == torch.float64
    for dt, tol in ((torch.float32, 1e-5), (torch.float16, 2e-2)):
        x = a.to(dt).requires_grad_(True)
        y = b.to(dt).requires_grad_(True)
        out = mod(x, y)
        assert out.dtype == dt and out.requires_grad
        w = torch.randn_like(out)
        gx, gy = torch.autograd.grad((out.float() * w.float()).sum(), [x, y])
        assert gx.dtype == dt and gy.dtype == dt
        xr, yr = x.detach().float().requires_grad_(True), y.detach().float().requires_grad_(True)
        rx, ry = torch.autograd.grad((_corr_reference(xr, yr, 3, 1, 3, 1, 1) * w.float()).sum(), [xr, yr])
        torch.testing.assert_close(gx.float(), rx, rtol=tol, atol=tol)
        torch.testing.assert_close(gy.float(), ry, rtol=tol, atol=tol)
    # double under grad (r4: the reference dispatches a double backward too, correlation_cuda_kernel.cu:495-541): gradients in
    # double, against torch's autograd of the double restatement at double tolerance
    x, y = a.double().requires_grad_(True), b.double().requires_grad_(True)
    out = mod(x, y)
    assert out.dtype == torch.float64 and out.requires_grad
    w = torch.randn_like(out)
    gx, gy = torch.autograd.grad((out * w).sum(), [x, y])
    xr, yr = x.detach().clone().requires_grad_(True), y.detach().clone().requires_grad_(True)
    rx, ry = torch.autograd.grad((_corr_reference(xr, yr, 3, 1, 3, 1, 1) * w).sum(), [xr, yr])
    assert gx.dtype == torch.float64 and gy.dtype == torch.float64
    torch.testing.assert_close(gx, rx, rtol=1e-12, atol=1e-13)
    torch.testing.assert_close(gy, ry, rtol=1e-12, atol=1e-13)
    with pytest.raises(RuntimeError, match="same dtype"):
        mod(a.double().requires_grad_(True), b)
    assert Correlation().stride2 == 2 and CorrelationFunction().max_displacement == 20  # the reference's defaults


def test_global_backward_skips_the_gradient_nobody_asked_for(ops):
    """a frozen reference frame (requires_grad False): only the query gradient is produced, and it equals the one of
    the both-sides backward"""
    torch.manual_seed(4)
    q = (torch.rand(9, 8, 16, device="cuda")).requires_grad_(True)
    k = torch.rand(9, 8, 16, device="cuda")
    lab = torch.randint(0, 2, (9, 8), dtype=torch.int32, device="cuda")
    out = ops.global_match(k, q, lab, 2)
    (gq,) = torch.autograd.grad(out.sum(), [q])
    k2 = k.clone().requires_grad_(True)
    q2 = q.detach().clone().requires_grad_(True)
    out2 = ops.global_match(k2, q2, lab, 2)
    gk2, gq2 = torch.autograd.grad(out2.sum(), [k2, q2])
    assert torch.equal(gq, gq2) and gk2.abs().sum() > 0


def _torch_knn_reference(ref, qry, lab, n_ids, k):
    """the reference's k > 1 formula (IntVOS.py:23-40, :76-94) restated with torch ops -- the CHECKER of the fuzz test below,
    differentiated by torch.autograd"""
    xs = (qry * qry).sum(1, keepdim=True)
    ys = (ref * ref).sum(1, keepdim=True).t()
    d = xs + ys - 2.0 * qry @ ref.t()                                     # [N, M]
    ids = torch.arange(n_ids, device=lab.device, dtype=lab.dtype)
    wrong = (lab[None, :] != ids[:, None]).float()                        # [n_ids, M]
    dd = d[:, None, :] + wrong[None] * 1e20                               # [N, n_ids, M]
    top = -torch.topk(-dd, k, dim=2).values
    valid = top < 1e20
    pad = (top * valid.float()).max(dim=2, keepdim=True).values
    return torch.where(valid, top, pad.expand_as(top)).mean(dim=2)        # [N, n_ids]


@pytest.mark.parametrize("seed", range(6))
def test_knn_gradients_fuzz_against_torch_autograd(ops, seed):
    """random shapes / label sets (objects with 0, 1, < k and many rows, unlabelled rows) for k = 2..8: forward values and both
    gradients of GlobalMatchTopkFn against torch.autograd through the formula above"""
    g = torch.Generator(device="cuda").manual_seed(100 + seed)
    C = [100, 16, 64, 7, 128, 33][seed]
    N, M = [257, 64, 500, 33, 300, 129][seed], [400, 90, 700, 40, 256, 513][seed]
    n_ids, k = [4, 3, 6, 2, 5, 3][seed], [3, 2, 8, 5, 4, 6][seed]
    ref = (torch.relu(torch.randn(M, C, generator=g, device="cuda")) * 0.3).requires_grad_(True)
    qry = (torch.relu(torch.randn(N, C, generator=g, device="cuda")) * 0.3).requires_grad_(True)
    lab = torch.randint(-1, max(n_ids - 2, 1), (M,), generator=g, device="cuda", dtype=torch.int32)
    if n_ids >= 3:
        lab[:2] = n_ids - 2   # an object with two rows (< k for k >= 3); the last id has none
    out = ops.global_match(ref, qry, lab, n_ids, k_nearest_neighbors=k)
    want = _torch_knn_reference(ref, qry, lab, n_ids, k)
    torch.testing.assert_close(out, want, rtol=2e-5, atol=2e-5)
    w = torch.randn(out.shape, generator=g, device="cuda")
    gr, gq = torch.autograd.grad((out * w).sum(), [ref, qry])
    wr, wq = torch.autograd.grad((want * w).sum(), [ref, qry])
    scale = float(wq.abs().max())
    torch.testing.assert_close(gq, wq, rtol=2e-4, atol=2e-5 * scale)
    torch.testing.assert_close(gr, wr, rtol=2e-4, atol=2e-5 * scale)


def _torch_local_full_reference(prev, cur, lab, n_ids, d):
    """IntVOS.py:299-313 + :398-432 restated with torch ops (the checker): raw full-resolution window distances with the 1e20
    padding, labels gathered at stride 2 with zero padding, where(mask, dist, 1.0), min over the window"""
    h, w, C = cur.shape
    P = 2 * d + 1
    ypad = torch.nn.functional.pad(prev, (0, 0, d, d, d, d), value=1e20)
    lpad = torch.nn.functional.pad(lab.float(), (2 * d, 2 * d, 2 * d, 2 * d))
    dists, labs = [], []
    for dy in range(P):
        for dx in range(P):
            dists.append(((cur - ypad[dy:dy + h, dx:dx + w]) ** 2).sum(2))
            labs.append(lpad[2 * dy:2 * dy + h, 2 * dx:2 * dx + w])
    dist, labs = torch.stack(dists, 2), torch.stack(labs, 2)                               # [h, w, P*P]
    ids = torch.arange(n_ids, device=cur.device).float()
    masked = torch.where(labs[..., None] == ids, dist[..., None], torch.ones_like(dist[..., None]))
    return masked.min(dim=2).values                                                        # [h, w, n_ids]


@pytest.mark.parametrize("seed", range(4))
def test_local_no_downsample_gradients_fuzz_against_torch_autograd(ops, seed):
    g = torch.Generator(device="cuda").manual_seed(200 + seed)
    C, h, w, d, n_ids = [(8, 9, 12, 1, 2), (16, 14, 11, 3, 3), (5, 20, 21, 2, 4), (32, 13, 17, 4, 2)][seed]
    prev = (torch.relu(torch.randn(C, h, w, generator=g, device="cuda")) * 0.5).requires_grad_(True)
    cur = (torch.relu(torch.randn(C, h, w, generator=g, device="cuda")) * 0.5).requires_grad_(True)
    lab = torch.randint(0, n_ids, (h, w), generator=g, device="cuda", dtype=torch.int32)
    out = ops.local_match(prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab, n_ids, d, downsample=False)
    want = _torch_local_full_reference(prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab, n_ids, d)
    torch.testing.assert_close(out, want, rtol=1e-5, atol=2e-6)
    wgt = torch.randn(out.shape, generator=g, device="cuda")
    gp, gc = torch.autograd.grad((out * wgt).sum(), [prev, cur])
    wp, wc = torch.autograd.grad((want * wgt).sum(), [prev, cur])
    scale = max(float(wc.abs().max()), 1e-6)
    torch.testing.assert_close(gp, wp, rtol=2e-4, atol=2e-5 * scale)
    torch.testing.assert_close(gc, wc, rtol=2e-4, atol=2e-5 * scale)
