"""On-device evaluation -- mirror of ``COOCC_Ray.evaluation_semantic`` (P/coocc/detectors/coocc_ray.py:659-684),
``fast_hist`` (:726-730) and ``cm_to_ious`` (P/utils/formating.py:4-14).

The reference resamples the logits to the ground-truth size, takes the argmax, copies prediction and
label to the host and runs ``np.bincount`` three times per prediction.  Here one kernel produces the
SC, SSC and visible-only SSC confusion matrices on the device; ``SemanticEvaluator`` accumulates them
over a whole validation set and reads them back once."""
import numpy as np
import torch

from ._lib import call, ptr

NOISE = 255


def _as_u8(t):
    if t.dtype != torch.uint8:
        t = t.to(torch.uint8)   # labels 0..C-1 and 255 survive the narrowing (astype(np.int) upstream)
    return t.contiguous()


def semantic_histograms(pred, gt, visible_mask=None, empty_idx=0, out=None, accumulate=False):
    """pred [1,C,h,w,d] float logits (any strides), gt [1,H,W,D] labels, visible_mask [1,H,W,D] or None
    -> int64 device tensor [4 + 2*C*C] = SC 2x2 | SSC CxC | OCC CxC, each [label][pred]."""
    if not pred.is_cuda:
        raise RuntimeError("co_occ_amd.evaluation runs on the HIP device only")
    assert pred.dim() == 5 and pred.shape[0] == 1 and gt.dim() == 4 and gt.shape[0] == 1, "batch size 1 (coocc_ray.py:662)"
    pred = pred.float()
    C, h, w, d = pred.shape[1:]
    H, W, D = gt.shape[1:]
    g = _as_u8(gt[0])
    v = _as_u8(visible_mask[0] != 0) if visible_mask is not None else None
    if out is None:
        out = torch.empty(4 + 2 * C * C, dtype=torch.int64, device=pred.device)
        accumulate = False
    sc, sx, sy, sz = pred.stride()[1:]
    call("coocc_eval_semantic", ptr(pred, strided=True), sc, sx, sy, sz, C, h, w, d, ptr(g), ptr(v) if v is not None else None,
         H, W, D, int(empty_idx), 1 if accumulate else 0, ptr(out))
    return out


def split_histograms(hist, C):
    return hist[:4].view(2, 2), hist[4:4 + C * C].view(C, C), hist[4 + C * C:].view(C, C)


def evaluation_semantic(pred, gt, eval_type, visible_mask=None, empty_idx=0):
    """Signature and return convention of coocc_ray.py:659: 'SC' -> (hist 2x2, None); 'SSC' -> (hist CxC,
    hist_occ CxC or None).  Histograms are int64 device tensors (``.cpu().numpy()`` gives the upstream arrays)."""
    C = pred.shape[1]
    sc, ssc, occ = split_histograms(semantic_histograms(pred, gt, visible_mask, empty_idx), C)
    if eval_type == 'SC':
        return sc, None
    if eval_type == 'SSC':
        return ssc, (occ if visible_mask is not None else None)
    raise ValueError("eval_type must be 'SC' or 'SSC'")


def cm_to_ious(cm):
    """formating.py:4-14: per-class tp / (pred + gt - tp) of a [label][pred] confusion matrix."""
    cm = np.asarray(cm, dtype=np.float64)
    tp = np.diag(cm)
    with np.errstate(divide="ignore", invalid="ignore"):
        return list(tp / (cm.sum(0) + cm.sum(1) - tp))


class SemanticEvaluator:
    """Whole-dataset accumulation on the device: ``update`` enqueues one kernel and never synchronises;
    ``compute`` does the single device->host copy."""

    def __init__(self, num_classes=17, empty_idx=0, device="cuda"):
        self.C, self.empty_idx = num_classes, empty_idx
        self.hist = torch.zeros(4 + 2 * num_classes * num_classes, dtype=torch.int64, device=device)

    def update(self, pred, gt, visible_mask=None):
        assert pred.shape[1] == self.C
        semantic_histograms(pred, gt, visible_mask, self.empty_idx, out=self.hist, accumulate=True)

    def compute(self):
        sc, ssc, occ = (t.cpu().numpy() for t in split_histograms(self.hist, self.C))
        ious = cm_to_ious(ssc)
        return dict(SC_metric=sc, SSC_metric=ssc, SSC_occ_metric=occ, SC_IoU=cm_to_ious(sc)[1],
                    SSC_mIoU=float(np.nanmean(ious[1:])), class_ious=ious)
