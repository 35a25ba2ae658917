"""Config defaults for the GLASS inference path.

`get_cfg()` restates the subset of detectron2 v0.6 `config/defaults.py` that the
inference path reads [d2-recall: detectron2 is not vendored in the reference and not
installed here]; the `add_*` functions carry the same names, keys and default values as
reference glass/config.py:10-214 so that the reference's predictor set-up sequence
(glass/inference/glass_runner.py:31-39) works unchanged against this package.
"""
from __future__ import annotations

from .cfgnode import CfgNode as CN

_CHARSET = ('0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ'
            '!"#$%&\'()*+,-./:;<=>?@[\\]^_`{|}~ ')

_D2_DEFAULTS = {
    "VERSION": 2,
    "VIS_PERIOD": 0,
    "SEED": -1,
    "OUTPUT_DIR": "./output",
    "MODEL": {
        "LOAD_PROPOSALS": False, "MASK_ON": False, "KEYPOINT_ON": False,
        "DEVICE": "cuda", "META_ARCHITECTURE": "GeneralizedRCNN", "WEIGHTS": "",
        "PIXEL_MEAN": [103.530, 116.280, 123.675], "PIXEL_STD": [1.0, 1.0, 1.0],
        "BACKBONE": {"NAME": "build_resnet_backbone", "FREEZE_AT": 2},
        "FPN": {"IN_FEATURES": [], "OUT_CHANNELS": 256, "NORM": "", "FUSE_TYPE": "sum"},
        "PROPOSAL_GENERATOR": {"NAME": "RPN", "MIN_SIZE": 0},
        "ANCHOR_GENERATOR": {
            "NAME": "DefaultAnchorGenerator", "SIZES": [[32, 64, 128, 256, 512]],
            "ASPECT_RATIOS": [[0.5, 1.0, 2.0]], "ANGLES": [[-90, 0, 90]], "OFFSET": 0.0},
        "RPN": {
            "HEAD_NAME": "StandardRPNHead", "IN_FEATURES": ["res4"], "BOUNDARY_THRESH": -1,
            "IOU_THRESHOLDS": [0.3, 0.7], "IOU_LABELS": [0, -1, 1], "BATCH_SIZE_PER_IMAGE": 256,
            "POSITIVE_FRACTION": 0.5, "BBOX_REG_LOSS_TYPE": "smooth_l1",
            "BBOX_REG_LOSS_WEIGHT": 1.0, "BBOX_REG_WEIGHTS": (1.0, 1.0, 1.0, 1.0),
            "SMOOTH_L1_BETA": 0.0, "LOSS_WEIGHT": 1.0, "PRE_NMS_TOPK_TRAIN": 12000,
            "PRE_NMS_TOPK_TEST": 6000, "POST_NMS_TOPK_TRAIN": 2000, "POST_NMS_TOPK_TEST": 1000,
            "NMS_THRESH": 0.7, "CONV_DIMS": [-1]},
        "ROI_HEADS": {
            "NAME": "Res5ROIHeads", "NUM_CLASSES": 80, "IN_FEATURES": ["res4"],
            "IOU_THRESHOLDS": [0.5], "IOU_LABELS": [0, 1], "BATCH_SIZE_PER_IMAGE": 512,
            "POSITIVE_FRACTION": 0.25, "SCORE_THRESH_TEST": 0.05, "NMS_THRESH_TEST": 0.5,
            "PROPOSAL_APPEND_GT": True},
        "ROI_BOX_HEAD": {
            "NAME": "", "BBOX_REG_LOSS_TYPE": "smooth_l1", "BBOX_REG_LOSS_WEIGHT": 1.0,
            "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0), "SMOOTH_L1_BETA": 0.0,
            "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 0, "POOLER_TYPE": "ROIAlignV2",
            "NUM_FC": 0, "FC_DIM": 1024, "NUM_CONV": 0, "CONV_DIM": 256, "NORM": "",
            "CLS_AGNOSTIC_BBOX_REG": False, "TRAIN_ON_PRED_BOXES": False},
        "ROI_MASK_HEAD": {
            "NAME": "MaskRCNNConvUpsampleHead", "POOLER_RESOLUTION": 14,
            "POOLER_SAMPLING_RATIO": 0, "NUM_CONV": 0, "CONV_DIM": 256, "NORM": "",
            "CLS_AGNOSTIC_MASK": False, "POOLER_TYPE": "ROIAlignV2"},
        "RESNETS": {
            "DEPTH": 50, "OUT_FEATURES": ["res4"], "NUM_GROUPS": 1, "NORM": "FrozenBN",
            "WIDTH_PER_GROUP": 64, "STRIDE_IN_1X1": True, "RES5_DILATION": 1,
            "RES2_OUT_CHANNELS": 256, "STEM_OUT_CHANNELS": 64,
            "DEFORM_ON_PER_STAGE": [False, False, False, False]},
    },
    "INPUT": {
        "MIN_SIZE_TRAIN": (800,), "MIN_SIZE_TRAIN_SAMPLING": "choice", "MAX_SIZE_TRAIN": 1333,
        "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333, "FORMAT": "BGR", "MASK_FORMAT": "polygon"},
    "DATASETS": {"TRAIN": (), "TEST": ()},
    "DATALOADER": {"NUM_WORKERS": 4, "ASPECT_RATIO_GROUPING": True},
    "SOLVER": {"IMS_PER_BATCH": 16},
    "TEST": {"EVAL_PERIOD": 0, "DETECTIONS_PER_IMAGE": 100},
}


def get_cfg() -> CN:
    """Work-alike of `detectron2.config.get_cfg()` for the keys the inference path reads."""
    return CN(_D2_DEFAULTS).clone()


def add_dataset_config(cfg: CN) -> None:
    """Same keys/values as reference glass/config.py:10-17 (training-side; carried only so
    reference YAML dumps merge cleanly)."""
    cfg.DATASETS.merge_from_other_cfg({
        "AUG": False, "RANDOM_CROP_PROB": 0.0, "IGNORE_DIFFICULT": False, "FIX_CROP": False,
        "CROP_SIZE": (512, 512), "MAX_ROTATE_THETA": 30, "FIX_ROTATE": False})


def add_glass_config(cfg: CN) -> None:
    """Same keys/values as reference glass/config.py:20-75."""
    cfg.MODEL.merge_from_other_cfg({
        "ROTATED_BOXES_ON": False, "ORIENTATION_ON": False,
        "ROI_HYBRID_HEAD": {"NAME": "ResBlockHybridHead", "POOLER_RESOLUTION": 64,
                            "NUM_FEATURES": 256, "DEPTH": 3, "NORM_IMG_CROPS": False},
        "FILTERED_RPN": {"IGNORE_TEXT": ["###", ""]},
        "LOCAL_FEATURE_EXTRACTOR": {"NAME": "ResNet_FeatureExtractor", "NUM_FEATURES": 256},
        "HYBRID_FUSION": {"NAME": "MultiAspectGCAttention", "NUM_FEATURES": 256, "RATIO": 0.5,
                          "HEADERS": 8, "FUSION_TYPE": "channel_add"},
        "ROI_ORIENTATION_HEAD": {"LOSS_WEIGHT": 1.0, "APPLY_TO_BOXES": False,
                                 "APPLY_TO_BOXES_DURING_TRAINING": True},
    })
    cfg.MODEL.ROI_MASK_HEAD.LOSS_WEIGHT = 0.005
    cfg.MODEL.ROI_HEADS.CLASS_NAMES = ["word"]
    cfg.INPUT.MIN_SIZE_TEST = 1600
    cfg.INPUT.MAX_SIZE_TEST = 1600
    cfg.INPUT.MAX_UPSCALE_RATIO = 2
    cfg.INPUT.merge_from_other_cfg({"ROTATION": {"ENABLED": False, "ANGLES": [0]}})
    cfg.TEST.IOU_THRESHOLD = 0.5
    cfg.TEST.USE_FILTERED_METRICS = True
    cfg.TEST.DONT_CARE_GT_LABELS = ["###", ""]


def _recognizer_block(name, backbone, encoder, decoder):
    return {
        "SAMPLE_WORDS_STRATEGY": "random", "SAMPLE_WORDS_STRATEGY_PROB": 0.3, "CLASS_IND": 0,
        "IGNORE_EMPTY_TEXT": True, "LABELS_TYPE": "attention", "MAX_WORD_LENGTH": 50,
        "CHARACTER_SET": _CHARSET, "UNK_SYMBOL_PRED": False, "POOLER_RESOLUTION_WIDTH": 32,
        "POOLER_RESOLUTION_HEIGHT": 32, "IN_FEATURES": ["p2", "p3", "p4", "p5", "p6"],
        "PAD_SAMPLER": "", "MAX_BATCH_SIZE": 256, "LOSS_WEIGHT": 2.0, "IGNORE_TEXT": ["###"],
        "SENSITIVE": True,
        "RECOGNIZER_HEAD": {
            "POOLER_PAD": {"NAME": ""},
            "BACKBONE": {"NAME": backbone},
            "ENCODER": {"NAME": encoder, "NUM_OF_LAYERS": 2, "HEIGHT_REDUCTION": "mean",
                        "N_HEAD": 8},
            "DECODER": {"NAME": decoder, "POS_ENC_HEIGHT_WIDTH": None}},
        **({"NAME": name} if name is not None else {}),
    }


def add_e2e_config(cfg: CN) -> None:
    """Same keys/values as reference glass/config.py:78-170."""
    cfg.MODEL.RECOGNIZER_ON = False
    # MI355X build only (not a reference key): "fp32" = the reference's arithmetic; "fp16" = convolutions / linears with
    # operands rounded to fp16 on the fp16 matrix cores, fp32 accumulate and storage (BASELINE.json configs[4])
    cfg.MODEL.CONV_PRECISION = "fp32"
    cfg.MODEL.ROI_MASK_HEAD.merge_from_other_cfg(
        _recognizer_block(None, "CNN_V1", "BiLSTMBlock", "ASTER"))
    cfg.MODEL.ROI_MASK_HEAD.MASK_INFERENCE = False
    blk = _recognizer_block("", "CNN_V1_2", "BiLSTMBlockV2", "ASTER_V2")
    blk.update({"POOLER_TYPE": "ROIAlignRotated", "NORM": "BN", "POOLER_SAMPLING_RATIO": 0,
                "CONV_DIM": 256, "SAMPLING_RATIO": 0})
    cfg.MODEL.merge_from_other_cfg({"ROI_RECOGNIZER_HEAD": blk})


def add_post_process_config(cfg: CN) -> None:
    """Same keys/values as reference glass/config.py:173-214."""
    cfg.merge_from_other_cfg({"POST_PROCESSING": {
        "NAME": "PostProcessorAcademic", "SKIP_ALL": False, "BOX_INFLATE_RATIO": 0.05,
        "BOX_PX_PADDING": [0, 0, 0, 0], "MIN_BOX_DIMENSION": 2, "MAX_OUTSIDE_BOX_MARGIN_PX": 5,
        "MERGE_IOA_THRESH": 0.3, "OVERLAP_WIDTH_PER_HEIGHT_THRESH": 0.3,
        "PAIRS_HEIGHT_RATIO_THRESH": 0.35, "LOW_CONFIDENCE": 0.01, "VALID_CONFIDENCE": 0.15,
        "DETECT_THRESHOLD": 0.25, "TEXT_THRESHOLD": 0.25, "MAX_ANGLE_DIFF": 15}})


def get_glass_cfg(config_path: str = None, opts=None) -> CN:
    """The reference predictor's cfg set-up (glass/inference/glass_runner.py:31-39) in one call."""
    cfg = get_cfg()
    add_e2e_config(cfg)
    add_glass_config(cfg)
    add_dataset_config(cfg)
    add_post_process_config(cfg)
    if config_path:
        cfg.merge_from_file(config_path)
    cfg.merge_from_list(list(opts or []))
    return cfg
