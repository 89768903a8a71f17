"""CPU: register / scratch budget of the hot kernels (VERDICT r5 "next" #5).  A spill or a lost occupancy step does not fail any
numerics test -- it only shows up as time on the GPU box; this test reads the compiler's own resource report
(`hipcc -Rpass-analysis=kernel-resource-usage`, device code only, no GPU needed) for every kernel of csrc/*.hip and pins

  * no private segment (scratch) and no VGPR spill in any kernel a propagated frame or the benchmark launches,
  * at most 256 registers per lane where two waves per SIMD are the design point, and the occupancy the kernels were tuned at.

The flags are the Makefile's (tools/kernel_resources.py prints the same table for a human)."""
import concurrent.futures
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "cvpr2020_manet_amd", "csrc")
FILES = ["global_match.hip", "local_match.hip", "seg_head.hip", "mask_step.hip", "correlation.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c"]


def _report(src):
    err = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, src), "-o", "/dev/null"], capture_output=True,
                         text=True, timeout=900).stderr
    rows, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark: +(.*?): (\d+) \[-Rpass", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    return rows


@pytest.fixture(scope="module")
def resources():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        reports = list(ex.map(_report, FILES))
    mangled = {}
    for r in reports:
        mangled.update(r)
    names = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
    out = {}
    for k, n in zip(mangled, names):
        n = re.sub(r"\(anonymous namespace\)::|^void ", "", n)
        out[re.sub(r"\(.*", "", n)] = mangled[k]
    assert len(out) > 80, "the resource report lists %d kernels: did the remark format change?" % len(out)
    return out


def _get(res, name):
    hits = [k for k in res if k == name]
    assert hits, "kernel %r is not in the report (renamed?): %s" % (name, sorted(res)[:400])
    return res[hits[0]]


# kernel -> (max VGPR + AGPR per lane, min waves per SIMD) -- every entry also means: no scratch, no VGPR spill
HOT = {
    # the headline fp32 global match at C = 100 (13 k-groups: KS = 50) and its arg-min twin; the bf16 kernel of configs[2] / [4]
    "global_match_f32_pipe_kernel<50, false>": (256, 2),
    "global_match_f32_pipe_kernel<50, true>": (256, 2),
    "global_match_bf16_wide_kernel<7, 0, false>": (256, 2),
    "global_finish_kernel": (128, 4),
    # per-frame operands (the staging forms 480p / 720p clips take)
    "frame_prepare_kernel<float, 32, true>": (128, 4),
    "frame_prepare_kernel<unsigned short, 32, true>": (128, 4),
    # local window: the fused kernel and its two halves (stored volumes, r6), the reference's window radius and configs[2]'s
    "local_fused_kernel<12, 0>": (256, 2), "local_fused_kernel<12, 1>": (256, 2), "local_fused_kernel<12, 2>": (128, 4),
    "local_fused_kernel<4, 0>": (128, 4), "local_fused_kernel<4, 1>": (128, 4), "local_fused_kernel<4, 2>": (128, 4),
    # the head
    "dwconv7x7_bn_relu_kernel<true, false, true, 0>": (128, 4),
    "dwconv7x7_bn_relu_kernel<false, false, true, 0>": (168, 3),
    "conv1x1_rw_kernel<64>": (256, 2), "conv1x1_rw_kernel<32>": (256, 2),
    "conv1x1_mfma_kernel": (168, 3),
    "head_layer1_object_kernel<1>": (128, 4), "head_layer1_object_kernel<2>": (128, 4),
    "relu_conv1x1_c1_kernel": (128, 4),
    "conv1x1_x3_kernel<0, 2>": (168, 3), "conv1x1_x3_kernel<0, 3>": (256, 2),
    # the mask step
    "upsample_argmax_kernel": (128, 4), "frame_begin_kernel": (128, 4),
}


def test_hot_kernels_keep_their_register_budget_and_touch_no_scratch(resources):
    bad = []
    for name, (regs, occ) in HOT.items():
        r = _get(resources, name)
        used = r.get("VGPRs", 0) + r.get("AGPRs", 0)
        if r.get("ScratchSize [bytes/lane]", 0) or r.get("VGPRs Spill", 0) or used > regs or r.get("Occupancy [waves/SIMD]", 0) < occ:
            bad.append((name, r))
    assert not bad, "register / scratch budget broken:\n" + "\n".join("%s: %s" % b for b in bad)


def test_every_window_radius_of_the_local_kernels_is_spill_free(resources):
    """d = 0..12 x (fused, volume out, volume in): VGPR spills never; d = 7..9 overflow into accumulation registers (a known cost:
    one wave per SIMD there) but stay out of scratch"""
    for d in range(13):
        for mode in range(3):
            r = _get(resources, "local_fused_kernel<%d, %d>" % (d, mode))
            assert r.get("VGPRs Spill", 0) == 0 and r.get("ScratchSize [bytes/lane]", 0) == 0, (d, mode, r)
            assert r.get("VGPRs", 0) <= 256


def test_no_kernel_of_the_library_spills_vector_registers_except_the_known_one(resources):
    """Anything else that starts spilling shows up here.  Known and accepted: the bf16r filter's arg variant (one VGPR, 8 bytes:
    global_match_bf16_wide_kernel<7, 0, true>, off the timed paths of the headline) and the SGPR-spill bookkeeping of the
    generic-stride fp32 frame prepare (36 bytes reserved, no scratch instruction in its code; odd widths / strided views only)."""
    known = {"global_match_bf16_wide_kernel<7, 0, true>", "frame_prepare_kernel<float, 32, false>"}
    bad = {k: v for k, v in resources.items()
           if (v.get("VGPRs Spill", 0) or v.get("ScratchSize [bytes/lane]", 0)) and k not in known}
    assert not bad, bad
    for k in known:
        assert _get(resources, k).get("ScratchSize [bytes/lane]", 0) <= 36
