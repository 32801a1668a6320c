"""Online tracking of one video sequence: the frame loop of the reference's ``Submitter.run``
(submit_engine.py:58-120) and its result writers (:133-184), without the dataset / logger plumbing.

    tracker = SequenceTracker(model, dataset_name="DanceTrack", det_score_thresh=0.5, ...)
    for frame_idx, (image, (ori_h, ori_w)) in enumerate(frames):      # image: (3,H,W) normalised tensor
        result = tracker.step(image, ori_h, ori_w)                     # filtered TrackInstances on the CPU
        lines += tracker.mot_lines(frame_idx, result)                  # "frame,id,x,y,w,h,1,-1,-1,-1"

One frame of lookahead (a recorded sequence always has it): ``tracker.step(image, h, w, next_image=frames[i + 1])``
queues the backbone + encoder of the NEXT frame -- the half of a frame that does not depend on the tracks, ~3/4 of
its kernel time -- on a side stream before the host blocks on this frame's scores, so the GPU works through it while
the host does the track bookkeeping and issues the query updater (results are identical; tests/test_model_gpu.py).
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
        from .utils.host import respect_cpu_quota
        respect_cpu_quota()           # (a container's CFS quota vs torch's machine-sized thread pool: utils/host.py)
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
        self._pending = None          # (image, encode result, event): the next frame's encode half, queued ahead
        self._slot = 0                # encode calls alternate between two graph slots (one may still be read)
        self._side = None

    @classmethod
    def from_config(cls, model, config: dict) -> "SequenceTracker":
        return cls(model, dataset_name=config["DATASET"], det_score_thresh=config["DET_SCORE_THRESH"],
                   track_score_thresh=config["TRACK_SCORE_THRESH"], result_score_thresh=config["RESULT_SCORE_THRESH"],
                   miss_tolerance=config["MISS_TOLERANCE"], use_dab=config["USE_DAB"])

    @torch.no_grad()
    def step(self, image: torch.Tensor, ori_h: int, ori_w: int, next_image: torch.Tensor = None) -> TrackInstances:
        """One frame: model -> runtime tracker -> query updater; returns the reportable tracks (CPU, boxes as
        xyxy pixels of the original image, low-score / tiny boxes removed).  ``next_image``: the frame the next call
        will pass (the same tensor object), whose encode half is then queued ahead on a side stream."""
        enc = self._encoded(image)
        res = self.model(tracks=self.tracks, encoded=enc)        # decoder + heads on this frame's encode result
        if next_image is not None:
            self._prefetch(next_image)
        previous, new = self.tracker.update(model_outputs=res, tracks=self.tracks)
        self.tracks = self.core.postprocess_single_frame(previous, new, None)
        return self._report(self.tracks[0], ori_h, ori_w)

    def _report(self, t: TrackInstances, ori_h: int, ori_w: int) -> TrackInstances:
        """The reportable tracks on the CPU (submit_engine.py:95-112): score and area filters, xyxy pixel boxes.
        The fields a result needs (ids, boxes, scores, labels) leave the device as ONE packed tensor through a
        pinned buffer and one event wait -- `.to("cpu")` field by field is a blocking copy per field, embeddings
        included."""
        n, K = len(t), t.scores.shape[-1] if t.scores.dim() == 2 else 0
        if n and t.boxes.is_cuda:
            packed = torch.cat((t.boxes.double(), t.scores.double().reshape(n, K), t.ids.double()[:, None],
                                t.labels.double()[:, None]), dim=1)
            host = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
            host.copy_(packed, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            done.synchronize()
        elif n:
            host = torch.cat((t.boxes.double(), t.scores.double().reshape(n, K), t.ids.double()[:, None],
                              t.labels.double()[:, None]), dim=1)
        else:
            host = torch.zeros((0, 6 + K), dtype=torch.float64)
        boxes, scores = host[:, :4].float(), host[:, 4:4 + K].float()
        ids, labels = host[:, 4 + K].long(), host[:, 5 + K].long()
        area = boxes[:, 2] * ori_w * boxes[:, 3] * ori_h
        keep = torch.ones((host.shape[0],), dtype=torch.bool)
        if host.shape[0]:
            keep = torch.max(scores, dim=-1).values > self.result_score_thresh
            keep = keep & (area > self.area_thresh)
        out = TrackInstances(hidden_dim=t.hidden_dim, num_classes=t.num_classes, use_dab=self.use_dab)
        out.ids, out.labels, out.scores, out.area = ids[keep], labels[keep], scores[keep], area[keep]
        out.boxes = box_cxcywh_to_xyxy(boxes[keep]) * torch.as_tensor([ori_w, ori_h, ori_w, ori_h], dtype=torch.float)
        return out

    def _encode(self, image: torch.Tensor) -> dict:
        frame = tensor_list_to_nested_tensor([image]).to(self.device)
        frame.encode_slot = self._slot          # (models/infer_graphs.py: one static `memory` per slot)
        frame.encode_static_ok = True           # ... read in place: this loop alternates the two slots itself
        self._slot ^= 1
        return self.model(frame=frame, stage="encode")

    def _encoded(self, image: torch.Tensor) -> dict:
        """This frame's encode result: the one queued ahead by the previous step, or computed now."""
        pending, self._pending = self._pending, None
        if pending is not None and pending[0] is image:
            torch.cuda.current_stream(self.device).wait_event(pending[2])
            return pending[1]
        if pending is not None:                 # a different frame arrived: nothing may still write the slot buffers
            torch.cuda.current_stream(self.device).wait_event(pending[2])
        return self._encode(image)

    def _prefetch(self, image: torch.Tensor) -> None:
        if self.device.type != "cuda":
            return
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        side = self._side
        side.wait_stream(main)                  # the image (and this frame's decode reading the OTHER slot) are on main
        with torch.cuda.stream(side):
            enc = self._encode(image)
            event = side.record_event()
        if image.is_cuda:
            image.record_stream(side)
        for v in enc.values():                  # produced on the side stream, consumed on main
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(main)
        self._pending = (image, enc, event)

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
