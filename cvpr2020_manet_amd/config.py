"""The flag namespace the drop-in module consults (`cfg`), host side.

It carries the same flag names, types and defaults as the reference's `config.py:17-80` so that either
this object or the reference's own `cfg` can be handed to `IntVOS(cfg, ...)`.  Differences: the flags are
declared in a table, unknown command-line flags are ignored (`parse_known_args`), and importing the
module does NOT require a GPU -- the reference raises at import without CUDA (`config.py:82-83`); here the
check lives where device work starts (`cvpr2020_manet_amd.ops` raises on CPU tensors), so host logic stays
testable.

Flags that steer the matching path: TEST_MODE, KNNS, MODEL_MAX_LOCAL_DISTANCE,
MODEL_LOCAL_DOWNSAMPLE, MODEL_SEMANTIC_EMBEDDING_DIM.
"""
import argparse
import os


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = str(v).strip().lower()
    if s in {"yes", "true", "t", "y", "1"}:
        return True
    if s in {"no", "false", "f", "n", "0"}:
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


B, I, F, S = str2bool, int, float, str

# (flag, type, default) -- grouped as in the reference
FLAGS = [
    # general
    ("ROOT_DIR", S, os.path.abspath(".")), ("EXP_NAME", S, "deeplabv3+coco"),
    ("SAVE_RESULT_DIR", S, "../afs/result/"), ("SAVE_VOS_RESULT_DIR", S, ""), ("NUM_WORKER", I, 1),
    ("KNNS", I, 1), ("PRETRAINED_MODEL", S, "./model_best.pth.tar"),
    ("RESULT_ROOT", S, "../afs/vos_result/result_total_80000"),
    # data
    ("DATA_NAME", S, "COCO2017"), ("DATA_AUG", B, True), ("DATA_WORKERS", I, 4), ("DATA_RESCALE", I, 416),
    ("DATA_RANDOMCROP", I, 416), ("DATA_RANDOMROTATION", I, 0), ("DATA_RANDOM_H", I, 10),
    ("DATA_RANDOM_S", I, 10), ("DATA_RANDOM_V", I, 10), ("DATA_RANDOMFLIP", F, 0.5),
    ("DATA_ROOT", S, "../data/DAVIS"),
    # model
    ("MODEL_NAME", S, "deeplabv3plus"), ("MODEL_BACKBONE", S, "res101_atrous"), ("MODEL_OUTPUT_STRIDE", I, 16),
    ("MODEL_ASPP_OUTDIM", I, 256), ("MODEL_SHORTCUT_DIM", I, 48), ("MODEL_SHORTCUT_KERNEL", I, 1),
    ("MODEL_NUM_CLASSES", I, 21), ("MODEL_SEMANTIC_EMBEDDING_DIM", I, 100), ("MODEL_HEAD_EMBEDDING_DIM", I, 256),
    ("MODEL_LOCAL_DOWNSAMPLE", B, True), ("MODEL_MAX_LOCAL_DISTANCE", I, 12), ("MODEL_SELECT_PERCENT", F, 0.8),
    ("MODEL_USEIntSeg", B, False),
    # MI355X matching path (not in the reference; a reference cfg without them gets the defaults):
    #   MODEL_MATCH_COMPUTE  arithmetic of the global match: f32 (exact) | bf16 | bf16x3 | bf16r (bf16 filter + fp32 re-rank)
    #   MODEL_EMB_DTYPE      storage of extract_feature's output in HBM: f32 | bf16 (2-byte embeddings end to end)
    #   MODEL_HEAD_POINTWISE the heads' 256-channel 1x1 layers in inference: f32 (exact fp32 MFMA) | split (split-bf16) | split3 (three-piece split: fp32-class) | framework
    #   MODEL_CACHE_FRAMES   keep prepared per-frame operands keyed on tensor identity (False: embeddings are rewritten in place)
    #   MODEL_LOCAL_VOLUME_CACHE_MB  cap of the stored local-match volumes (IntVOS.prepare_local_volumes; LRU beyond it)
    #   MODEL_LOCAL_VOLUME_LAZY      store a frame pair's volume at its first use (no prepare_local_volumes call needed)
    #   MODEL_HEAD_MEMO_MB           cap of the heads' memoised layer-1 shared-half terms on the cached frames
    ("MODEL_MATCH_COMPUTE", S, "f32"), ("MODEL_EMB_DTYPE", S, "f32"), ("MODEL_HEAD_POINTWISE", S, "f32"),
    ("MODEL_CACHE_FRAMES", B, True), ("MODEL_LOCAL_VOLUME_CACHE_MB", I, 8192), ("MODEL_LOCAL_VOLUME_LAZY", B, False),
    ("MODEL_HEAD_MEMO_MB", I, 8192),
    # train
    ("TRAIN_LR", F, 0.0007), ("TRAIN_LR_GAMMA", F, 0.1), ("TRAIN_MOMENTUM", F, 0.9),
    ("TRAIN_WEIGHT_DECAY", F, 0.00004), ("TRAIN_POWER", F, 0.9), ("TRAIN_BATCH_SIZE", I, 2),
    ("TRAIN_SHUFFLE", B, True), ("TRAIN_CLIP_GRAD_NORM", F, 5.0), ("TRAIN_MINEPOCH", I, 9),
    ("TRAIN_TOTAL_STEPS", I, 101000), ("TRAIN_LOSS_LAMBDA", I, 0), ("TRAIN_TBLOG", B, False),
    ("TRAIN_BN_MOM", F, 0.0003), ("TRAIN_TOP_K_PERCENT_PIXELS", F, 0.15), ("TRAIN_HARD_MINING_STEP", I, 50000),
    ("TRAIN_LR_STEPSIZE", I, 2000), ("TRAIN_INTER_USE_TRUE_RESULT", B, True), ("TRAIN_RESUME_DIR", S, ""),
    # logging / test
    ("LOG_DIR", S, "./log"), ("TEST_CHECKPOINT", S, "save_step_100000.pth"), ("TEST_MODE", B, False),
]


def build_parser():
    parser = argparse.ArgumentParser(description="intvos config", add_help=False)
    for name, kind, default in FLAGS:
        parser.add_argument("--" + name, type=kind, default=default)
    return parser


def make_cfg(argv=None):
    """cfg namespace from an argv list (None: the process's sys.argv; unknown flags are ignored)."""
    ns, _ = build_parser().parse_known_args(argv)
    ns.TRAIN_EPOCHS = int(200000 * ns.TRAIN_BATCH_SIZE / 60.0)  # derived, as config.py:80
    return ns


cfg = make_cfg()
