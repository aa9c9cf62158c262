"""Proposal files -> ``bounding_boxes`` of an instant-ngp transforms.json (reference nerf_rpn/scripts/proposals2ngp.py:10-196).
A consumer of the hot path's output files (``<save_path>/proposals/<scene>.npz``: ``proposal`` [K,6|7], ``score`` [K]); host-side numpy
only -- K <= 30 boxes per scene.  Grid coordinates -> scene coordinates through the feature file's bbox_min/bbox_max/resolution, z-up ->
y-up, then instant-ngp's matrix convention (axis cycle or Mitsuba flip, y/z sign flip, (t - offset) / scale)."""
import argparse
import json
import os

import numpy as np

_Z_UP_TO_Y_UP = np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]])


def ngp_matrix_to_nerf(ngp_matrix, scale, offset, from_mitsuba):
    m = np.array(ngp_matrix, dtype=np.float64, copy=True)
    if from_mitsuba:
        m[:, [0, 2]] *= -1
    else:
        m = m[[2, 0, 1], :]              # cycle axes xyz -> yzx
    m[:, [1, 2]] *= -1
    m[:, 3] = (m[:, 3] - offset) / scale
    return m


def _boxes(rotations, centres, extents, scale, offset, from_mitsuba):
    offset = _Z_UP_TO_Y_UP @ offset
    out = []
    for rot, centre, ext in zip(rotations, centres, extents):
        xform = _Z_UP_TO_Y_UP @ np.concatenate((rot, centre[:, None]), axis=1)
        xform = ngp_matrix_to_nerf(xform, scale, offset, from_mitsuba)
        out.append({"orientation": xform[:3, :3].tolist(), "position": xform[:3, 3].tolist(), "extents": np.asarray(ext).tolist()})
    return out


def proposals_to_ngp_boxes(proposals, features_dict):
    """Axis-aligned proposals [K,6] in grid coordinates."""
    res, lo, hi = features_dict["resolution"], features_dict["bbox_min"], features_dict["bbox_max"]
    scale, diag = features_dict["scale"], features_dict["bbox_max"] - features_dict["bbox_min"]
    bmin, bmax = proposals[:, :3] / res * diag + lo, proposals[:, 3:] / res * diag + lo
    return _boxes([np.eye(3)] * len(bmin), (bmin + bmax) * 0.5, (bmax - bmin) / scale, scale, features_dict["offset"],
                  features_dict["from_mitsuba"])


def obb_to_ngp_boxes(proposals, features_dict):
    """Rotated proposals [K,7] = (x, y, z, w, l, h, theta) in grid coordinates."""
    res, lo = features_dict["resolution"], features_dict["bbox_min"]
    scale, diag = features_dict["scale"], features_dict["bbox_max"] - features_dict["bbox_min"]
    pos, ext = proposals[:, :3] / res * diag + lo, proposals[:, 3:6] / res * diag / scale
    rots = [np.array([[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1]]) for t in proposals[:, 6]]
    return _boxes(rots, pos, ext, scale, features_dict["offset"], features_dict["from_mitsuba"])


def process_scene(args, proposal_path, json_path, feature_path, output_path):
    assert os.path.isfile(json_path) and os.path.isfile(feature_path)
    with open(json_path) as f:
        json_dict = json.load(f)
    p, feats = np.load(proposal_path), np.load(feature_path)
    scores, proposals = p["score"], p["proposal"]
    keep = scores > args.threshold
    scores, proposals = scores[keep], proposals[keep]
    order = np.argsort(scores)[::-1]
    scores, proposals = scores[order][:args.top_k], proposals[order][:args.top_k]
    print(f"{os.path.basename(proposal_path).split('.')[0]}: {len(scores)} proposals")
    boxes = proposals_to_ngp_boxes(proposals, feats) if args.bbox_format == "aabb" else obb_to_ngp_boxes(proposals, feats)
    for b, s in zip(boxes, scores):
        b["score"] = s.item()
    json_dict["bounding_boxes"] = boxes
    with open(output_path, "w") as f:
        json.dump(json_dict, f, indent=2)


def build_parser():
    p = argparse.ArgumentParser(description="Convert the RPN proposals to bounding boxes in instant-ngp transforms.json.")
    p.add_argument("--bbox_format", choices=["aabb", "obb"], required=True)
    p.add_argument("--dataset", type=str, required=True, choices=["hypersim", "front3d"], help="Dataset name. Must be hypersim or front3d.")
    p.add_argument("--dataset_path", default="", help="Path to the NeRF scenes.")
    p.add_argument("--features_path", default="", help="Path to the NeRF features.")
    p.add_argument("--proposals_path", default="", help="Path to the proposal files.")
    p.add_argument("--output_dir", default="", help="Path to the output directory.")
    p.add_argument("--threshold", default=0.5, type=float, help="The threshold for the proposal scores.")
    p.add_argument("--top_k", default=30, type=int, help="The number of proposals to visualize.")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    os.makedirs(args.output_dir, exist_ok=True)
    for f in sorted(os.listdir(args.proposals_path)):
        if f.endswith(".npz") and os.path.isfile(os.path.join(args.proposals_path, f)):
            name = f.split(".")[0]
            process_scene(args, os.path.join(args.proposals_path, f), os.path.join(args.dataset_path, name, "train", "transforms.json"),
                          os.path.join(args.features_path, name + ".npz"), os.path.join(args.output_dir, name + ".json"))


if __name__ == "__main__":
    main()
