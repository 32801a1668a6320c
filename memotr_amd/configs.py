"""Model / training hyper-parameters of the reference's shipped configs, as python dicts.

Values transcribed from configs/train_dancetrack.yaml, train_mot17.yaml and train_bdd100k.yaml of the
reference (flat UPPER_CASE keys, read by the ``build(config)`` functions exactly like the reference's).
Only the keys the per-frame path and the train step read are kept; data-pipeline keys are out of scope.
"""
from __future__ import annotations


def dancetrack_config(**overrides) -> dict:
    cfg = dict(
        MODE="train", VISUALIZE=False, AVAILABLE_GPUS="0,1,2,3,4,5,6,7", DEVICE="cuda", USE_DISTRIBUTED=False,
        USE_CHECKPOINT=False, CHECKPOINT_LEVEL=2, DATASET="DanceTrack", BATCH_SIZE=1, ACCUMULATION_STEPS=1,
        # model (configs/train_dancetrack.yaml:57-75)
        BACKBONE="resnet50", HIDDEN_DIM=256, FFN_DIM=2048, NUM_FEATURE_LEVELS=4, NUM_HEADS=8, NUM_ENC_POINTS=4,
        NUM_DEC_POINTS=4, NUM_ENC_LAYERS=6, NUM_DEC_LAYERS=6, MERGE_DET_TRACK_LAYER=1, ACTIVATION="ReLU",
        RETURN_INTER_DEC=True, EXTRA_TRACK_ATTN=False, AUX_LOSS=True, USE_DAB=True, UPDATE_THRESH=0.5,
        LONG_MEMORY_LAMBDA=0.01,
        # sampling / training (:78-101)
        SAMPLE_STEPS=[6, 10, 14], SAMPLE_LENGTHS=[2, 3, 4, 5], SEED=42, EPOCHS=20,
        ONLY_TRAIN_QUERY_UPDATER_AFTER=20, DROPOUT=0.0, NUM_DET_QUERIES=300, TP_DROP_RATE=0.0, FP_INSERT_RATE=0.0,
        LR=2.0e-4, LR_BACKBONE=2.0e-5, LR_POINTS=1.0e-5, WEIGHT_DECAY=0.0005, CLIP_MAX_NORM=0.1,
        LR_SCHEDULER="MultiStep", LR_DROP_RATE=0.1, LR_DROP_MILESTONES=[12],
        # matcher / loss (:103-111)
        MATCH_COST_CLASS=2, MATCH_COST_BBOX=5, MATCH_COST_GIOU=2, LOSS_WEIGHT_FOCAL=2, LOSS_WEIGHT_L1=5,
        LOSS_WEIGHT_GIOU=2, AUX_LOSS_WEIGHT=[1.0, 1.0, 1.0, 1.0, 1.0],
        # inference thresholds (:24-33)
        DET_SCORE_THRESH=0.5, TRACK_SCORE_THRESH=0.5, RESULT_SCORE_THRESH=0.5, MISS_TOLERANCE=30, USE_MOTION=False,
    )
    cfg.update(overrides)
    return cfg


def mot17_config(**overrides) -> dict:
    """configs/train_mot17.yaml: same model; clips of up to 4 frames (:80), MOT17 + CrowdHuman joint training."""
    cfg = dancetrack_config(DATASET="MOT17", SAMPLE_LENGTHS=[2, 3, 4], MISS_TOLERANCE=15)
    cfg.update(overrides)
    return cfg


def bdd100k_config(**overrides) -> dict:
    """configs/train_bdd100k.yaml: 8 classes, 720x1280 inputs, clips of up to 4 frames (:67), MISS_TOLERANCE 10 (:26)."""
    cfg = dancetrack_config(DATASET="BDD100K", SAMPLE_LENGTHS=[2, 3, 4], MISS_TOLERANCE=10)
    cfg.update(overrides)
    return cfg
