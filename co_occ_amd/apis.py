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


def pipelined_test(model, data_iter, slots=6, dense_streams=3, ahead=0, stats=None):
    """The loop of ``custom_single_gpu_test`` (P/coocc/apis/test.py:43-45: ``result = model(return_loss=False, **data)`` once per
    sample) with ``slots`` samples in flight (``co_occ_amd.serving``): a generator of ``(data, result)`` in sample order, ``result``
    = what ``COOCC_Ray.simple_test`` returns for that sample (same tensors, same metrics, fine outputs trimmed to their exact
    size).  ``data``: the keyword arguments of ``simple_test`` (``img_inputs`` / ``img``, ``points``, ``gt_occ``,
    ``visible_mask``, ``precomputed``).

    Nothing in the loop waits for the sample that was just issued: the encoders upstream of the hot path run eagerly at submit
    time; pooling + index search of the next ``ahead`` samples are prefetched under the dense stages (one captured hipGraph
    launch per sample) of the current ones; a sample's SC / SSC confusion matrices (``coocc_eval_semantic``) and its fine-point
    count are computed on ITS dense stream right behind the replay and copied to pinned host memory asynchronously; the sample
    is yielded ``dense_streams`` issues later, when that copy has normally long finished.  A sample whose ``gt_occ`` is not the
    captured fine grid (cascade_ratio x the coarse grid) takes ``model.simple_test`` (eager decode) in its turn.  Result tensors
    of a sample stay valid until ``slots`` more samples have been submitted: consume (or clone) them inside the loop body, as
    the upstream loop does.  ``stats`` (optional dict) receives ``fallbacks`` / ``recaptures`` / ``eager_samples``."""
    import collections
    model.eval()
    pipe = None
    submitted = collections.deque()        # (data, ticket | None, eager result | None): search dispatched, dense not issued
    issued = collections.deque()           # (data, ticket, pinned host buffer, event): dense stage + metrics enqueued
    ncls_of = lambda out: out["pred_c"].shape[1]
    eager = 0

    def issue(item):
        """Dense stage of the oldest submitted sample onto its stream (no host wait) + metrics + the async host copy."""
        data, t, res = item
        if t is None:
            return (data, None, res, None)
        out = t.result(wait=False)
        ds = pipe.dense_streams[t.slot % pipe.ndense]
        gt, vm = data.get("gt_occ"), data.get("visible_mask")
        with torch.cuda.stream(ds):
            parts = []
            if gt is not None:
                gt.record_stream(ds)
                if vm is not None:
                    vm.record_stream(ds)
                both = model._metrics_launch(out, gt, vm)
                parts.append(both.reshape(-1))
            if out.get("fine_count") is not None:
                parts.append(out["fine_count"].reshape(-1).to(torch.int64))
            host = ev = None
            if parts and not (model.metrics_on_device and gt is not None and len(parts) == 1):
                dev_buf = torch.cat(parts) if len(parts) > 1 else parts[0]
                host = torch.empty(dev_buf.shape, dtype=torch.int64, pin_memory=True)
                host.copy_(dev_buf, non_blocking=True)
            from . import streams as cstreams
            ev = cstreams.new_event()           # fires after the pinned-host copy above: a copy command, complete when it does
            ev.record(ds)
        return (data, t, (out, host, both if gt is not None else None), ev)

    def finish(item):
        data, t, payload, ev = item
        if t is None:
            return data, payload
        out, host, both = payload
        ev.synchronize()
        from . import core
        core.check_h2_overflow()
        out = dict(out)
        gt, vm = data.get("gt_occ"), data.get("visible_mask")
        C = ncls_of(out)
        nm = 0
        if gt is not None:
            nm = both.numel()
        if out.get("fine_count") is not None and out.get("output_voxels_fine") is not None and not t.fallback:
            cf = model.pts_bbox_head.cascade_ratio
            n = int(host[nm]) * cf ** 3
            # capacity-sized fine outputs of the captured form -> the exact-size tensors simple_test returns (views of the
            # slot's static buffers: no copy; the device-count kernels pack [3][n] at the start of the coords buffer)
            out["output_voxels_fine"] = [out["output_voxels_fine"][0][:n]]
            out["output_coords_fine"] = [out["output_coords_fine"][0].reshape(-1)[:3 * n].view(3, n)]
        out.update(output_voxels=out["pred_c"], target_voxels=gt)
        if gt is not None:
            m = both if model.metrics_on_device else host[:nm].numpy().reshape(both.shape).copy()
            out.update(model._metrics_finish(m, C, vm is not None))
        return data, out

    try:
        with torch.no_grad():
            for data in data_iter:
                img = data.get("img_inputs", data.get("img"))
                fr = model.serving_frame(img=img, points=data.get("points"), img_metas=data.get("img_metas"),
                                         precomputed=data.get("precomputed"))
                if pipe is None:
                    pipe = model.serving(fr, slots=slots, dense_streams=dense_streams, ahead=ahead,
                                         render=bool(model.use_rendering and model.test_rendering))
                gt = data.get("gt_occ")
                X, Y, Z = pipe.grid
                cf = model.pts_bbox_head.cascade_ratio
                if gt is not None and list(gt.shape[1:]) != [X * cf, Y * cf, Z * cf]:
                    # the captured scatter writes the default fine grid: this sample goes through simple_test's eager decode
                    keep, model.graph_simple_test = model.graph_simple_test, False
                    try:
                        res = model.simple_test(**{("img" if k == "img_inputs" else k): v for k, v in data.items()})
                    finally:
                        model.graph_simple_test = keep
                    eager += 1
                    submitted.append((data, None, res))
                else:
                    submitted.append((data, pipe.submit(fr), None))
                while len(submitted) > pipe.ahead:
                    issued.append(issue(submitted.popleft()))
                while len(issued) > pipe.ndense:
                    yield finish(issued.popleft())
            while submitted:
                issued.append(issue(submitted.popleft()))
                while len(issued) > pipe.ndense:
                    yield finish(issued.popleft())
            while issued:
                yield finish(issued.popleft())
    finally:
        if pipe is not None:
            if stats is not None:
                stats.update(fallbacks=pipe.fallbacks, recaptures=pipe.recaptures, eager_samples=eager)
            pipe.drain()
            pipe.close()
