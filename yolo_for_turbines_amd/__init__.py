"""yolo_for_turbines_amd — MI355X-native YOLOv3 hot path (forward, box decode, NMS) behind the
call surface of GabeTsai/YOLO-For-Turbines (`code/model.py`, `code/utils.py`).

The on-disk directory is also reachable as ``yolo-for-turbines_amd`` (symlink; a hyphen is not a
legal Python identifier). Everything computes in ``libyolo_mi355x.so`` (hand-written gfx950 HIP);
there is no CPU fallback.
"""
from .graph import GraphedTrainStep
from .loss import FusedYOLOLoss, YOLOLoss
from .optim import SGD
from .model import CNNBlock, ResidualBlock, ScalePredictionBlock, YOLOv3, layer_config
from .utils import (save_checkpoint, load_checkpoint, accuracy_counts, build_targets, calc_iou, calc_mAP, check_model_accuracy, eval_boxes, get_eval_boxes, letterbox, unletterbox_boxes, cells_to_boxes, decode_boxes, detect, detect_images, iou_aligned, nms_indices,
                    non_max_suppression)

__all__ = ["YOLOLoss", "FusedYOLOLoss", "GraphedTrainStep", "SGD", "CNNBlock", "ResidualBlock", "ScalePredictionBlock", "YOLOv3", "layer_config", "calc_iou",
           "cells_to_boxes", "decode_boxes", "detect", "detect_images", "iou_aligned", "nms_indices", "non_max_suppression", "build_targets", "calc_mAP", "accuracy_counts", "check_model_accuracy", "eval_boxes", "get_eval_boxes", "letterbox", "unletterbox_boxes"]
