"""Prediction dump of the nuScenes test loop -- mirror of ``save_output_nuscenes`` (P/coocc/apis/utils.py:54-110) and of the
label preparation in ``custom_single_gpu_test`` (P/coocc/apis/test.py:67-68,197-201).

Upstream resamples the logits to the ground-truth grid with ``F.interpolate``, takes ``argmax`` on the device, copies an
int64 volume to the host and narrows it to uint8 there.  ``predict_labels`` produces the uint8 volume on the device in one
kernel (no resampled [1,C,H,W,D] temporary: 174 MB at 17 x 200 x 200 x 16 fp32); the pickle written by
``save_output_nuscenes`` has the upstream keys (``pred_voxels``, ``cam2lidar``, ``img_canvas``)."""
import os
import pickle

import numpy as np
import torch

from ._lib import call, ptr

CAMERA_NAMES = ['CAM_FRONT_LEFT', 'CAM_FRONT', 'CAM_FRONT_RIGHT', 'CAM_BACK_LEFT', 'CAM_BACK', 'CAM_BACK_RIGHT']


def predict_labels(pred, size=None):
    """pred [1,C,h,w,d] float logits (any strides) -> uint8 device tensor [1,H,W,D] = argmax over classes of the logits
    trilinearly resampled to ``size`` (default: the logits' own grid)."""
    if not pred.is_cuda:
        raise RuntimeError("co_occ_amd.apis.predict_labels runs on the HIP device only")
    assert pred.dim() == 5 and pred.shape[0] == 1, "batch size 1 (test.py:59)"
    pred = pred.float()
    C, h, w, d = pred.shape[1:]
    H, W, D = [int(v) for v in (size if size is not None else (h, w, d))]
    out = torch.empty(1, H, W, D, dtype=torch.uint8, device=pred.device)
    sc, sx, sy, sz = pred.stride()[1:]
    call("coocc_predict_labels", ptr(pred, strided=True), sc, sx, sy, sz, C, h, w, d, H, W, D, ptr(out))
    return out


def save_output_nuscenes(img_inputs, output_voxels, save_path, scene_token, sample_token, img_filenames, timestamp,
                         scene_name):
    """Same arguments and file as upstream.  ``output_voxels``: [1,X,Y,Z] labels (``predict_labels`` output, or the
    upstream int64 argmax).  ``img_filenames``: {camera name: path} or None (then ``img_canvas`` is an empty list)."""
    rots, trans = img_inputs[1:3]
    num_img = rots.shape[1]
    cam2lidar = np.repeat(np.eye(4)[np.newaxis], repeats=num_img, axis=0)
    cam2lidar[:, :3, :3] = rots[0].detach().cpu().numpy()
    cam2lidar[:, :3, -1] = trans[0].detach().cpu().numpy()
    labels = output_voxels[0].detach().cpu().numpy().astype(np.uint8)
    canvas = []
    if img_filenames:
        from PIL import Image
        for name in CAMERA_NAMES:
            canvas.append(Image.open(img_filenames[name]).resize([480, 270], Image.BILINEAR))
    out_dict = dict(pred_voxels=labels, cam2lidar=cam2lidar, img_canvas=canvas)
    if scene_name is not None:
        save_path = os.path.join(save_path, scene_name)
        filepath = os.path.join(save_path, str(timestamp) + '.pkl')
    else:
        filepath = os.path.join(save_path, sample_token + '.pkl')
    os.makedirs(save_path, exist_ok=True)
    with open(filepath, "wb") as handle:
        pickle.dump(out_dict, handle)
    return filepath


def pipelined_test(model, data_iter, slots=6, dense_streams=3, ahead=0):
    """The loop of ``custom_single_gpu_test`` (P/coocc/apis/test.py:43-45: ``result = model(return_loss=False, **data)`` once per
    sample) with ``slots`` samples in flight (``co_occ_amd.serving``): a generator of ``(data, result)`` in sample order, ``result``
    = what ``COOCC_Ray.simple_test`` returns for that sample (same tensors, same metrics).  ``data``: the keyword arguments of
    ``simple_test`` (``img_inputs`` / ``img``, ``points``, ``gt_occ``, ``visible_mask``, ``precomputed``).  The encoders upstream
    of the hot path run eagerly at submit time; pooling + index search of the next samples are prefetched under the dense stage
    (one captured hipGraph launch per sample) of the current ones.  Result tensors of a sample stay valid until ``slots`` more
    samples have been submitted: consume (or clone) them inside the loop body, as the upstream loop does."""
    import collections
    model.eval()
    pipe = None
    inflight = collections.deque()

    def finish(item):
        data, t = item
        out = t.result(wait=True)
        return data, model.finish_test_result(out, data.get("gt_occ"), data.get("visible_mask"))

    with torch.no_grad():
        for data in data_iter:
            img = data.get("img_inputs", data.get("img"))
            fr = model.serving_frame(img=img, points=data.get("points"), img_metas=data.get("img_metas"),
                                     precomputed=data.get("precomputed"))
            if pipe is None:
                pipe = model.serving(fr, slots=slots, dense_streams=dense_streams, ahead=ahead,
                                     render=bool(model.use_rendering and model.test_rendering))
            inflight.append((data, pipe.submit(fr)))
            while len(inflight) > pipe.ahead:
                yield finish(inflight.popleft())
        while inflight:
            yield finish(inflight.popleft())
    if pipe is not None:
        pipe.close()
