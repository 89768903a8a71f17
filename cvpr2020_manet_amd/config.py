"""The reference's global `cfg` namespace (reference config.py:17-80), host side.

Same flag names, types and defaults as the reference, parsed from sys.argv with
`parse_known_args` so that importing this module never fails on foreign flags.  Unlike the
reference (config.py:82-83) importing does NOT require a GPU: the check moved to where device
work starts (cvpr2020_manet_amd.ops raises on CPU tensors), so that host logic is testable.

Flags that steer the matching path: TEST_MODE, KNNS, MODEL_MAX_LOCAL_DISTANCE,
MODEL_LOCAL_DOWNSAMPLE, MODEL_SEMANTIC_EMBEDDING_DIM.
"""
import argparse
import os


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def build_parser():
    p = argparse.ArgumentParser(description="intvos config", add_help=False)
    a = p.add_argument
    a("--ROOT_DIR", type=str, default=os.path.abspath("."))
    a("--EXP_NAME", type=str, default="deeplabv3+coco")
    a("--SAVE_RESULT_DIR", type=str, default="../afs/result/")
    a("--SAVE_VOS_RESULT_DIR", type=str, default="")
    a("--NUM_WORKER", type=int, default=1)
    a("--KNNS", type=int, default=1)
    a("--PRETRAINED_MODEL", type=str, default="./model_best.pth.tar")
    a("--RESULT_ROOT", type=str, default=os.path.join("../afs/vos_result/result_total_80000"))
    # data
    a("--DATA_NAME", type=str, default="COCO2017")
    a("--DATA_AUG", type=str2bool, default=True)
    a("--DATA_WORKERS", type=int, default=4)
    a("--DATA_RESCALE", type=int, default=416)
    a("--DATA_RANDOMCROP", type=int, default=416)
    a("--DATA_RANDOMROTATION", type=int, default=0)
    a("--DATA_RANDOM_H", type=int, default=10)
    a("--DATA_RANDOM_S", type=int, default=10)
    a("--DATA_RANDOM_V", type=int, default=10)
    a("--DATA_RANDOMFLIP", type=float, default=0.5)
    a("--DATA_ROOT", type=str, default="../data/DAVIS")
    # model
    a("--MODEL_NAME", type=str, default="deeplabv3plus")
    a("--MODEL_BACKBONE", type=str, default="res101_atrous")
    a("--MODEL_OUTPUT_STRIDE", type=int, default=16)
    a("--MODEL_ASPP_OUTDIM", type=int, default=256)
    a("--MODEL_SHORTCUT_DIM", type=int, default=48)
    a("--MODEL_SHORTCUT_KERNEL", type=int, default=1)
    a("--MODEL_NUM_CLASSES", type=int, default=21)
    a("--MODEL_SEMANTIC_EMBEDDING_DIM", type=int, default=100)
    a("--MODEL_HEAD_EMBEDDING_DIM", type=int, default=256)
    a("--MODEL_LOCAL_DOWNSAMPLE", type=str2bool, default=True)
    a("--MODEL_MAX_LOCAL_DISTANCE", type=int, default=12)
    a("--MODEL_SELECT_PERCENT", type=float, default=0.8)
    a("--MODEL_USEIntSeg", type=str2bool, default=False)
    # train
    a("--TRAIN_LR", type=float, default=0.0007)
    a("--TRAIN_LR_GAMMA", type=float, default=0.1)
    a("--TRAIN_MOMENTUM", type=float, default=0.9)
    a("--TRAIN_WEIGHT_DECAY", type=float, default=0.00004)
    a("--TRAIN_POWER", type=float, default=0.9)
    a("--TRAIN_BATCH_SIZE", type=int, default=2)
    a("--TRAIN_SHUFFLE", type=str2bool, default=True)
    a("--TRAIN_CLIP_GRAD_NORM", type=float, default=5.0)
    a("--TRAIN_MINEPOCH", type=int, default=9)
    a("--TRAIN_TOTAL_STEPS", type=int, default=101000)
    a("--TRAIN_LOSS_LAMBDA", type=int, default=0)
    a("--TRAIN_TBLOG", type=str2bool, default=False)
    a("--TRAIN_BN_MOM", type=float, default=0.0003)
    a("--TRAIN_TOP_K_PERCENT_PIXELS", type=float, default=0.15)
    a("--TRAIN_HARD_MINING_STEP", type=int, default=50000)
    a("--TRAIN_LR_STEPSIZE", type=int, default=2000)
    a("--TRAIN_INTER_USE_TRUE_RESULT", type=str2bool, default=True)
    a("--TRAIN_RESUME_DIR", type=str, default="")
    a("--LOG_DIR", type=str, default=os.path.join("./log"))
    a("--TEST_CHECKPOINT", type=str, default="save_step_100000.pth")
    a("--TEST_MODE", type=str2bool, default=False)
    return p


def make_cfg(argv=None):
    """cfg namespace from an argv list (None: the process's sys.argv, unknown flags ignored)."""
    ns, _ = build_parser().parse_known_args(argv)
    ns.TRAIN_EPOCHS = int(200000 * ns.TRAIN_BATCH_SIZE / 60.0)
    return ns


cfg = make_cfg()
