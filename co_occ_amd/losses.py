"""Occupancy losses of ``OccHead.loss`` (P/coocc/dense_heads/occ_head.py:265-337): class-weighted cross-entropy, the
semantic / geometric scene-class affinity losses (P/utils/semkitti.py:65-149) and Lovasz-softmax
(P/coocc/dense_heads/lovasz_softmax.py:156-203), plus the majority-vote label pooling of ``loss_voxel`` (:269-281).

Host-side eager torch on the logits the HIP path produced, exactly as upstream (the losses are a few reductions over
80 k coarse voxels / <= 8 x fine_topk fine points; they are downstream of the hot path).  Restated in
vectorised form -- one softmax, one one-hot matmul for all per-class sums, one column-wise sort for all present classes
-- instead of the reference's per-class Python loops; the arithmetic (sums, ratios, -log clamped at 100) is the same, so
values agree to float rounding (tests/golden/losses.npz, generated from the unmodified reference functions)."""
import numpy as np
import torch
import torch.nn.functional as F

# voxel counts per nuScenes-Occupancy class (data: P/utils/nusc_param.py:10-12), class_weights = 1 / log(freq + 1e-3)
NUSC_CLASS_FREQUENCIES = np.array([2242961742295, 25985376, 1561108, 28862014, 196106643, 15920504, 2158753, 26539491, 4004729,
                                   34838681, 75173306, 2255027978, 50959399, 646022466, 869055679, 1446141335, 1724391378])


def nusc_class_weights():
    return torch.from_numpy(1 / np.log(NUSC_CLASS_FREQUENCIES + 0.001))


def _bce_to_one(x):
    """F.binary_cross_entropy(x, ones) = -log(x) with torch's clamp of the log at -100.  Written with a substituted
    argument so that x == 0 gives the value 100 and a ZERO gradient (0 * inf would be NaN; torch's own backward clamps the
    denominator at 1e-12 instead -- a degenerate case, e.g. a present class that receives no probability mass at all)."""
    ok = x > 1e-43
    return torch.where(ok, -torch.log(torch.where(ok, x, torch.ones_like(x))), torch.full_like(x, 100.0))


def _flat(pred, target):
    """[B,C,...] logits + [B,...] labels -> ([P,C], [P])."""
    C = pred.shape[1]
    return pred.reshape(pred.shape[0], C, -1).permute(0, 2, 1).reshape(-1, C), target.reshape(-1)


def ce_ssc_loss(pred, target, class_weights=None, ignore_index=255):
    """semkitti.py:140-149."""
    return F.cross_entropy(pred, target.long(), weight=class_weights, ignore_index=ignore_index, reduction="mean")


def geo_scal_loss(pred, ssc_target, ignore_index=255, non_empty_idx=0):
    """semkitti.py:65-90: precision / recall / specificity of the occupied-vs-empty split, each through BCE against 1."""
    logits, t = _flat(pred, ssc_target)
    p = F.softmax(logits, dim=1)
    valid = t != ignore_index
    empty = p[:, non_empty_idx][valid]
    nonempty = 1 - empty
    tgt = (t != non_empty_idx)[valid].float()
    eps = 1e-5
    inter = (tgt * nonempty).sum()
    precision = inter / (nonempty.sum() + eps)
    recall = inter / (tgt.sum() + eps)
    spec = ((1 - tgt) * empty).sum() / ((1 - tgt).sum() + eps)
    return _bce_to_one(precision) + _bce_to_one(recall) + _bce_to_one(spec)


def sem_scal_loss(pred, ssc_target, ignore_index=255):
    """semkitti.py:93-137: for every class present in the target, -log precision (if any probability mass), -log recall,
    -log specificity (if any negative), averaged over the present classes."""
    logits, t = _flat(pred, ssc_target)
    valid = t != ignore_index
    p = F.softmax(logits, dim=1)[valid]                       # [P,C]
    t = t[valid].long()
    C = p.shape[1]
    onehot = F.one_hot(t.clamp(max=C - 1), C).to(p.dtype) * (t < C).unsqueeze(1)
    n_t = onehot.sum(0)                                       # voxels of class i
    nom = (p * onehot).sum(0)
    sum_p = p.sum(0)
    neg = p.shape[0] - n_t                                    # sum(1 - completion_target)
    spec_num = ((1 - p) * (1 - onehot)).sum(0)
    present = n_t > 0
    zero = torch.zeros_like(nom)
    l_prec = torch.where(sum_p > 0, _bce_to_one(nom / sum_p.clamp(min=1e-38)), zero)
    l_rec = _bce_to_one(nom / n_t.clamp(min=1))
    l_spec = torch.where(neg > 0, _bce_to_one(spec_num / neg.clamp(min=1)), zero)
    per_class = torch.where(present, l_prec + l_rec + l_spec, zero)
    return per_class.sum() / present.sum()


def lovasz_softmax(probas, labels, ignore=None):
    """lovasz_softmax.py:156-203 with classes='present', per_image=False: mean over present classes of
    <sorted errors, Lovasz gradient>; all present classes are sorted in one column-wise sort."""
    if probas.dim() > 2:
        probas, labels = _flat(probas, labels)
    if ignore is not None:
        valid = labels != ignore
        probas, labels = probas[valid], labels[valid]
    if probas.numel() == 0:
        return probas.sum() * 0.
    C = probas.shape[1]
    labels = labels.long()
    present = torch.bincount(labels.clamp(max=C), minlength=C + 1)[:C] > 0
    cls = torch.nonzero(present).flatten()
    fg = (labels.unsqueeze(1) == cls.unsqueeze(0)).to(probas.dtype)          # [P, Cp]
    errors = (fg - probas[:, cls]).abs()
    errors_sorted, perm = torch.sort(errors, 0, descending=True)
    fg_sorted = torch.gather(fg, 0, perm)
    gts = fg_sorted.sum(0, keepdim=True)
    inter = gts - fg_sorted.cumsum(0)
    union = gts + (1 - fg_sorted).cumsum(0)
    jac = 1. - inter / union
    jac = torch.cat([jac[:1], jac[1:] - jac[:-1]], 0)                        # lovasz_grad (:21-33)
    return (errors_sorted * jac).sum(0).mean()


def pool_labels(target_voxels, H, W, D, empty_idx=0, num_cls=17):
    """Label volume [B,rH,rW,rD] -> [B,H,W,D] by the majority vote of ``loss_voxel`` (occ_head.py:269-281): a cell whose
    children are all empty stays empty; otherwise the most frequent NON-empty label wins (smallest label on ties); when no
    non-empty label occurs twice the upstream trick (empty children get unique negative ids, torch.mode returns the
    smallest mode) yields 255 as soon as one child is empty and the smallest label otherwise."""
    B = target_voxels.shape[0]
    ratio = target_voxels.shape[1] // H
    if ratio == 1:
        return target_voxels.long()
    t = target_voxels.reshape(B, H, ratio, W, ratio, D, ratio).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, H, W, D, ratio ** 3).long()
    # labels are 0 .. num_cls-1 and 255 (ignore): 255 takes the extra bin num_cls, which keeps it the LARGEST label in the
    # "smallest label on ties" order (a [.., 256] histogram was 82 MB per step at 100x100x8)
    nlab = num_cls + 1
    counts = torch.zeros(B, H, W, D, nlab, dtype=torch.int32, device=t.device)
    counts.scatter_add_(-1, torch.where(t == 255, torch.full_like(t, num_cls), t.clamp(0, num_cls - 1)), torch.ones_like(t, dtype=torch.int32))
    n_empty = counts[..., empty_idx].clone()
    all_empty = t.sum(-1) == empty_idx
    counts[..., empty_idx] = 0
    m, arg = counts.max(-1)                                   # first (= smallest label) maximum
    arg = torch.where(arg == num_cls, torch.full_like(arg, 255), arg)
    out = torch.where((m == 1) & (n_empty > 0), torch.full_like(arg, 255), arg)
    return torch.where(all_empty, torch.full_like(arg, empty_idx), out).long()
