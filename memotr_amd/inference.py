"""Online tracking of one video sequence: the frame loop of the reference's ``Submitter.run``
(submit_engine.py:58-120) and its result writers (:133-184), without the dataset / logger plumbing.

    tracker = SequenceTracker(model, dataset_name="DanceTrack", det_score_thresh=0.5, ...)
    for frame_idx, (image, (ori_h, ori_w)) in enumerate(frames):      # image: (3,H,W) normalised tensor
        result = tracker.step(image, ori_h, ori_w)                     # filtered TrackInstances on the CPU
        lines += tracker.mot_lines(frame_idx, result)                  # "frame,id,x,y,w,h,1,-1,-1,-1"
"""
from __future__ import annotations

from typing import List

import torch

from .models.runtime_tracker import RuntimeTracker
from .models.utils import get_model
from .structures.track_instances import TrackInstances
from .utils.box_ops import box_cxcywh_to_xyxy
from .utils.nested_tensor import tensor_list_to_nested_tensor

BDD_CLS2LABEL = {1: "pedestrian", 2: "rider", 3: "car", 4: "truck", 5: "bus", 6: "train", 7: "motorcycle",
                 8: "bicycle"}
MOT_STYLE = ("DanceTrack", "SportsMOT", "MOT17", "MOT17_SPLIT")


class SequenceTracker:
    def __init__(self, model, dataset_name: str = "DanceTrack", det_score_thresh: float = 0.7,
                 track_score_thresh: float = 0.6, result_score_thresh: float = 0.7, miss_tolerance: int = 5,
                 use_dab: bool = True, area_thresh: int = 100):
        self.model = model.eval()
        self.core = get_model(model)
        self.dataset_name = dataset_name
        self.result_score_thresh = result_score_thresh
        self.area_thresh = area_thresh
        self.use_dab = use_dab
        self.device = next(self.core.parameters()).device
        self.tracker = RuntimeTracker(det_score_thresh=det_score_thresh, track_score_thresh=track_score_thresh,
                                      miss_tolerance=miss_tolerance, use_dab=use_dab)
        self.tracks: List[TrackInstances] = [TrackInstances(hidden_dim=self.core.hidden_dim,
                                                            num_classes=self.core.num_classes,
                                                            use_dab=use_dab).to(self.device)]

    @classmethod
    def from_config(cls, model, config: dict) -> "SequenceTracker":
        return cls(model, dataset_name=config["DATASET"], det_score_thresh=config["DET_SCORE_THRESH"],
                   track_score_thresh=config["TRACK_SCORE_THRESH"], result_score_thresh=config["RESULT_SCORE_THRESH"],
                   miss_tolerance=config["MISS_TOLERANCE"], use_dab=config["USE_DAB"])

    @torch.no_grad()
    def step(self, image: torch.Tensor, ori_h: int, ori_w: int) -> TrackInstances:
        """One frame: model -> runtime tracker -> query updater; returns the reportable tracks (CPU, boxes as
        xyxy pixels of the original image, low-score / tiny boxes removed)."""
        frame = tensor_list_to_nested_tensor([image]).to(self.device)
        res = self.model(frame=frame, tracks=self.tracks)
        previous, new = self.tracker.update(model_outputs=res, tracks=self.tracks)
        self.tracks = self.core.postprocess_single_frame(previous, new, None)
        out = self.tracks[0].to(torch.device("cpu"))
        out.area = out.boxes[:, 2] * ori_w * out.boxes[:, 3] * ori_h
        out = out[torch.max(out.scores, dim=-1).values > self.result_score_thresh] if len(out) else out
        out = out[out.area > self.area_thresh] if len(out) else out
        out.boxes = box_cxcywh_to_xyxy(out.boxes) * torch.as_tensor([ori_w, ori_h, ori_w, ori_h], dtype=torch.float)
        return out

    def mot_lines(self, frame_idx: int, tracks: TrackInstances) -> List[str]:
        if self.dataset_name not in MOT_STYLE:
            raise ValueError(f"{self.dataset_name} dataset is not supported for submit process.")
        lines = []
        for box, tid in zip(tracks.boxes.tolist(), tracks.ids.tolist()):
            x1, y1, x2, y2 = box
            lines.append(f"{frame_idx + 1},{tid},{x1},{y1},{x2 - x1},{y2 - y1},1,-1,-1,-1\n")
        return lines

    @staticmethod
    def bdd_frame_result(frame_idx: int, tracks: TrackInstances, img_path: str) -> dict:
        name = img_path.split("/")[-1]
        labels = []
        for box, tid, lab in zip(tracks.boxes.tolist(), tracks.ids.tolist(), tracks.labels.tolist()):
            x1, y1, x2, y2 = box
            labels.append({"id": str(tid), "category": BDD_CLS2LABEL[lab + 1],
                           "box2d": {"x1": x1, "y1": y1, "x2": x2, "y2": y2}})
        return {"name": name, "videoName": name[:-12], "frameIndex": frame_idx, "labels": labels}
