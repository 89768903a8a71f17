#!/usr/bin/env python3
"""GPU fuzz of the kernels around the path against torch: depthwise 7x7 + BN + ReLU, the fp32 / split 1x1
kernels (any Cin, add term, fused output layer), head input assembly, label resize, upsample + argmax -- random shapes.
usage: tools/fuzz_head.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from cvpr2020_manet_amd import ops  # noqa: E402


def run(cases, seed, verbose=False):
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())  # noqa: E731
    gd = torch.Generator(device=dev).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=gd, device=dev)  # noqa: E731
    bad = 0

    def check(name, got, want, rtol, atol, info):
        nonlocal bad
        ok = torch.allclose(got, want, rtol=rtol, atol=atol)
        if not ok:
            bad += 1
            print("BAD %s %s max|d| %.3g" % (name, info, float((got - want).abs().max())))
        elif verbose:
            print("ok  %s %s" % (name, info))

    with torch.no_grad():
        for _ in range(cases):
            kind = ri(0, 7)
            if kind == 0:  # depthwise 7x7 + BN + ReLU
                B, C, h, w = ri(1, 4), ri(1, 40), ri(1, 70), ri(1, 80)
                x = rn(B, C, h, w)
                wt, b = rn(C, 1, 7, 7) * 0.2, rn(C)
                sc, sh = rn(C).abs() + 0.5, rn(C)
                relu, relu_in = bool(ri(0, 1)), bool(ri(0, 1))
                want = F.conv2d(torch.relu(x) if relu_in else x, wt, b, padding=3, groups=C) * sc.view(1, C, 1, 1) + sh.view(1, C, 1, 1)
                want = torch.relu(want) if relu else want
                got = ops.dwconv7x7_bn_relu(x, wt, b, scale=sc, shift=sh, relu=relu, relu_in=relu_in)
                check("dwconv", got, want, 1e-4, 1e-4, (B, C, h, w, relu, relu_in))
            elif kind in (1, 2):  # 1x1 kernels
                B, cin = ri(1, 4), [1, 3, 4, 7, 16, 32, 33, 64, 100, 103, 128, 256][ri(0, 11)]
                h, w = ri(1, 40), 4 * ri(1, 20)
                x = rn(B, cin, h, w)
                w2t, b2 = rn(cin, 256) * 0.1, rn(256)
                want = F.conv2d(x, w2t.t().reshape(256, cin, 1, 1).contiguous(), b2)
                relu = bool(ri(0, 1))
                form = ri(0, 2)
                tol = 2e-4 if kind == 1 else 2e-3
                fn = ops.conv1x1_mfma if kind == 1 else (lambda x_, w_, b_, **k: ops.conv1x1_split(x_, ops.SplitWeight(w_), b_, **k))
                if form == 0:
                    got = fn(x, w2t, b2, relu_out=relu)
                    check("pw%d" % kind, got, torch.relu(want) if relu else want, tol, tol, (B, cin, h, w, relu))
                elif form == 1:
                    add = rn(256, h, w)
                    got = fn(x, w2t, b2, relu_out=relu, add=add)
                    ref = want + add
                    check("pw%d+add" % kind, got, torch.relu(ref) if relu else ref, tol, tol, (B, cin, h, w, relu))
                else:
                    hw_, hb_ = rn(1, 256, 1, 1) * 0.1, rn(1)
                    got = fn(x, w2t, b2, head_weight=hw_, head_bias=hb_)
                    check("pw%d+head" % kind, got, F.conv2d(torch.relu(want), hw_, hb_), tol, tol * 4, (B, cin, h, w))
            elif kind == 3:  # label resize (nearest) as F.interpolate(mask.float(), size, 'nearest').int()
                H, W, h, w = ri(4, 200), ri(4, 300), ri(1, 60), ri(1, 80)
                m = torch.randint(0, 6, (1, 1, H, W), generator=gd, device=dev)
                want = F.interpolate(m.float(), size=(h, w), mode="nearest").int()
                got = ops.label_resize_nearest(m, (h, w))
                check("label_resize", got.float().reshape(-1), want.float().reshape(-1), 0, 0, (H, W, h, w))
            elif kind == 4:  # upsample (bilinear, align_corners) + argmax
                n, h, w, H, W = ri(1, 6), ri(2, 40), ri(2, 50), ri(2, 160), ri(2, 200)
                lg = rn(1, n, h, w)
                up = F.interpolate(lg, size=(H, W), mode="bilinear", align_corners=True)
                got, _ = ops.upsample_argmax(lg, (H, W), want_small=False)
                # ties / near-ties may flip with the interpolation's rounding: compare the chosen logits, not the indices
                chosen = torch.gather(up, 1, got.long().view(1, 1, H, W)).view(-1)
                check("upsample_argmax", chosen, up.max(1).values.view(-1), 0, 2e-5, (n, h, w, H, W))
            elif kind == 6:  # r5: layer 1's per-object half in one launch against the torch chain on the assembled channels
                n_ids, h, w = ri(1, 6), ri(1, 70), 4 * ri(1, 35)
                gm, lm = torch.rand(h, w, n_ids, generator=gd, device=dev), torch.rand(h, w, n_ids, generator=gd, device=dev)
                lab = torch.randint(-1, n_ids + 1, (h, w), generator=gd, device=dev, dtype=torch.int32)
                wd, bd = rn(3, 1, 7, 7) * 0.2, (rn(3) if ri(0, 1) else None)
                sc, sh = rn(3).abs() + 0.5, rn(3)
                w2, b2, term = rn(3, 256) * 0.3, rn(256), rn(1, 256, h, w)
                relu = bool(ri(0, 1))
                got = ops.head_layer1_object(gm, lm, lab, n_ids, (h, w), wd, bd, sc, sh, w2, b2, term, relu_out=relu)
                ids = torch.arange(n_ids, device=dev).view(n_ids, 1, 1)
                x = torch.stack((gm.permute(2, 0, 1), lm.permute(2, 0, 1), (lab.unsqueeze(0) == ids).float()), 1)
                y = torch.relu(F.conv2d(x, wd, bd, padding=3, groups=3) * sc.view(1, 3, 1, 1) + sh.view(1, 3, 1, 1))
                want = F.conv2d(y, w2.t().reshape(256, 3, 1, 1).contiguous(), b2) + term
                check("layer1_object", got, torch.relu(want) if relu else want, 1e-4, 1e-4, (n_ids, h, w, relu))
            elif kind == 7:  # r5: the resident-weights 1x1 kernel on planes large enough for ranges cut in half-tile units
                B, cin = ri(1, 5), [32, 64, 96, 128, 256][ri(0, 4)]
                h, w = ri(50, 130), 4 * ri(20, 55)
                x = rn(B, cin, h, w)
                w2t, b2 = rn(cin, 256) * 0.1, rn(256)
                relu = bool(ri(0, 1))
                want = F.conv2d(x, w2t.t().reshape(256, cin, 1, 1).contiguous(), b2)
                got = ops.conv1x1_mfma(x, w2t, b2, relu_out=relu)
                check("pw_rw", got, torch.relu(want) if relu else want, 2e-4, 2e-4, (B, cin, h, w, relu))
                one = ops.conv1x1_mfma(x[B - 1:].contiguous(), w2t, b2, relu_out=relu)  # (cut elsewhere: the same bits)
                check("pw_rw cut", got[B - 1:], one, 0, 0, (B, cin, h, w, relu))
            else:  # head inputs
                n_ids, h, w = ri(1, 6), ri(1, 40), ri(1, 50)
                gm, lm = torch.rand(1, h, w, n_ids, 1, generator=gd, device=dev), torch.rand(1, h, w, n_ids, 1, generator=gd, device=dev)
                lab = torch.randint(-1, n_ids, (h, w, 1), generator=gd, device=dev, dtype=torch.int32)
                got = ops.head_inputs(gm, lm, lab, n_ids, (h, w))
                ids = torch.arange(n_ids, device=dev).float()
                want = torch.cat((gm.squeeze(0).permute(2, 3, 0, 1), lm.squeeze(0).permute(2, 3, 0, 1),
                                  (lab.float() == ids).unsqueeze(-1).permute(2, 3, 0, 1).float()), 1)
                check("head_inputs", got, want, 0, 0, (n_ids, h, w))
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    b = run(n, s, verbose=os.environ.get("VERBOSE") == "1")
    print("%d cases, %d mismatches" % (n, b))
    sys.exit(1 if b else 0)
