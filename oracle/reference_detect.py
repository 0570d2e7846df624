"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement (numpy, fp32 where the reference computes in fp32) of the reference's detection INFERENCE post-processing:
  * intersections / jaccard                     Applications/VisionModels/retinanet.py:500-521
  * nms (top_k sort, greedy same-class NMS, relative thresholds, single-inclusion filter, cross-class duplicate filter,
    max_boxes)                                   retinanet.py:523-711
  * BBoxPredictor.__call__ (threshold on the best class, box decode with mean/std, clip to the image, drop empty boxes, nms)
                                                 retinanet.py:713-812
  * mAP1 / mAP                                   Applications/Vision.py:1696-1800
  * ComputeMaxOverlaps                           Applications/Vision.py:1666-1694
Pinned by tests/golden/g11_bbox_inference.npz, generated from the real reference by oracle/gen_golden.py (g11).
"""
import numpy as np

F32 = np.float32


def intersections(B1, B2):
    "pairwise intersection areas [n,m] of min-max boxes (retinanet.py:500-509)"
    B1, B2 = np.asarray(B1)[:, None, :], np.asarray(B2)[None, :, :]
    w = np.clip(np.minimum(B1[..., 2], B2[..., 2]) - np.maximum(B1[..., 0], B2[..., 0]), 0, None)
    h = np.clip(np.minimum(B1[..., 3], B2[..., 3]) - np.maximum(B1[..., 1], B2[..., 1]), 0, None)
    return w * h


def jaccard(B1, B2):
    "pairwise IoU [n,m] (retinanet.py:511-521): inter / (a1 + a2 - inter)"
    B1, B2 = np.asarray(B1), np.asarray(B2)
    a1 = (B1[:, 2] - B1[:, 0]) * (B1[:, 3] - B1[:, 1])
    a2 = (B2[:, 2] - B2[:, 0]) * (B2[:, 3] - B2[:, 1])
    inter = intersections(B1, B2)
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter / (a1[:, None] + a2[None, :] - inter)


def _drop(lst, idxs):
    idxs = set(int(i) for i in idxs)
    return [v for i, v in enumerate(lst) if i not in idxs]


def nms(boxes, classes, scores, max_overlap=0.5, rel_thresh=None, top_k=1000, max_boxes=20, dup=None, inc=None):
    """retinanet.py:523-711.  boxes [N,4] fp32, classes [N] int, scores [N] fp32 -> (list of boxes, classes, scores), scores
    descending."""
    boxes, classes, scores = np.asarray(boxes, F32), np.asarray(classes), np.asarray(scores, F32)
    if len(boxes) == 0:
        return [], [], []
    order = np.argsort(-scores, kind='stable')[:top_k]              # :581-584 (descending sort, keep top_k)
    S, C, B = list(scores[order]), list(classes[order]), list(boxes[order])
    S2, C2, B2 = [], [], []
    while S:                                                         # :598-604 greedy NMS inside a class
        jac = jaccard(np.array(B)[:1], np.array(B))[0]
        kill = np.where((jac > max_overlap) & (np.array(C) == C[0]))[0]
        if 0 not in kill:          # a zero-area / NaN head never deletes itself: the reference loops forever here (BBoxPredictor
            raise ValueError('nms: degenerate box (IoU with itself is not > max_overlap)')   # removes empty boxes beforehand)
        S2.append(S[0]); C2.append(C[0]); B2.append(B[0])
        S, C, B = _drop(S, kill), _drop(C, kill), _drop(B, kill)
    S, C, B = S2, C2, B2
    if rel_thresh:                                                   # :613-635
        t1, t2 = rel_thresh
        for i in range(len(S)):
            if S[i] < t1 * S[0]:
                S, C, B = S[:i], C[:i], B[:i]
                break
        kill = []
        for i in range(len(S) - 1):
            for j in range(i + 1, len(S)):
                if C[i] == C[j] and S[j] < t2 * S[i]:
                    kill.append(j)
        S, C, B = _drop(S, kill), _drop(C, kill), _drop(B, kill)
    if inc:                                                          # :641-671 single inclusions of the same class
        thr, inc_classes = inc
        L = len(C)
        if L:
            pc, pb = np.array(C), np.array(B)
            eq = (pc[:, None] == pc[None, :]).astype(int)
            inter = intersections(pb, pb)
            area = (pb[:, 2] - pb[:, 0]) * (pb[:, 3] - pb[:, 1])
            ratios = inter / area                                    # [i,j] = |i ∩ j| / |j|   (numpy broadcast over the LAST axis)
            ratios2 = area[None, :] / area[:, None]
            incl = (ratios * eq > thr).astype(int) - np.identity(L, int)
            big = incl * (ratios2 > 0.25).astype(int)
            single = list((big.sum(axis=1) == 1).nonzero()[0])
            single = [i for i in single if int(C[i]) not in inc_classes]
            partners = [int(np.argmax(big[i])) for i in single]
            single = list(set(single) - set(partners))
            kill = []
            for i in single:
                j = int(np.argmax(big[i]))
                if S[i] < 0.75 * S[j]:
                    kill.append(i)
                elif S[j] < 0.75 * S[i]:
                    kill.append(j)
            S, C, B = _drop(S, kill), _drop(C, kill), _drop(B, kill)
    if dup:                                                          # :677-695 duplicates of different classes
        thr, pairs = dup
        again = True
        while again:
            again = False
            if len(B) == 0:
                break
            jac = jaccard(np.array(B), np.array(B))
            L = len(B)
            for i in range(L - 1):
                hit = -1
                for j in range(i + 1, L):
                    if jac[i, j] > thr and (C[i], C[j]) in pairs and S[j] < 0.75 * S[i]:
                        hit = j
                        break
                if hit >= 0:
                    S, C, B = S[:hit] + S[hit + 1:], C[:hit] + C[hit + 1:], B[:hit] + B[hit + 1:]
                    again = True
                    break
    return B[:max_boxes], C[:max_boxes], S[:max_boxes]


def bbox_predict(img_hw, reg, clas, anchors, thresh=0.05, max_overlap=0.5, rel_thresh=None, top_k=1000, max_boxes=20,
                 dup=None, inc=None, mean=(0., 0., 0., 0.), std=(0.1, 0.1, 0.2, 0.2)):
    """BBoxPredictor.__call__ (retinanet.py:733-812).  img_hw = (height, width); reg [bs,A,4], clas [bs,A,K], anchors [A,4]."""
    height, width = img_hw
    reg, clas, anchors = np.asarray(reg, F32), np.asarray(clas, F32), np.asarray(anchors, F32)
    mean, std = np.asarray(mean, F32), np.asarray(std, F32)
    W = anchors[:, 2] - anchors[:, 0]
    H = anchors[:, 3] - anchors[:, 1]
    Cx = anchors[:, 0] + F32(0.5) * W
    Cy = anchors[:, 1] + F32(0.5) * H
    PB, PC, CS = [], [], []
    for i in range(len(reg)):
        conf = clas[i].max(axis=1)
        cls = clas[i].argmax(axis=1)
        keep = np.nonzero(conf > F32(thresh))[0]
        if len(keep) == 0:
            PB.append([]); PC.append([]); CS.append([])
            continue
        conf, cls, r = conf[keep], cls[keep], reg[i][keep]
        w, h, cx, cy = W[keep], H[keep], Cx[keep], Cy[keep]
        dx, dy = r[:, 0] * std[0] + mean[0], r[:, 1] * std[1] + mean[1]
        dw, dh = r[:, 2] * std[2] + mean[2], r[:, 3] * std[3] + mean[3]
        pcx, pcy = cx + w * dx, cy + h * dy
        pw, ph = w * np.exp(dw), h * np.exp(dh)
        b = np.stack([pcx - F32(0.5) * pw, pcy - F32(0.5) * ph, pcx + F32(0.5) * pw, pcy + F32(0.5) * ph], 1).astype(F32)
        b[:, 0] = np.maximum(b[:, 0], 0); b[:, 1] = np.maximum(b[:, 1], 0)
        b[:, 2] = np.minimum(b[:, 2], F32(width)); b[:, 3] = np.minimum(b[:, 3], F32(height))
        good = np.nonzero(((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0))[0]
        bb, cc, ss = nms(b[good], cls[good], conf[good], max_overlap, rel_thresh, top_k, max_boxes, dup, inc)
        PB.append(bb); PC.append(cc); CS.append(ss)
    return PB, PC, CS


def mAP1(targs, preds, scores, thresh):
    "Vision.py:1696-1747: AP of one category at one IoU threshold (area under the max-smoothed precision curve)"
    N = len(targs)
    is_correct, all_scores = [], []
    for i in range(N):
        ok = [0] * len(preds[i])
        if len(preds[i]) > 0 and len(targs[i]) > 0:
            jac = jaccard(np.array(targs[i], F32), np.array(preds[i], F32))
            for j in range(jac.shape[0]):
                k = int(np.argmax(jac[j]))
                if jac[j, k] > thresh:
                    ok[k] = 1
        is_correct += ok
        all_scores += list(scores[i])
    combined = sorted(zip(all_scores, is_correct), reverse=True)
    ic = np.array([c for _, c in combined])
    L = len(ic)
    ntrue = sum(len(t) for t in targs)
    tp = np.cumsum(ic)
    precision = tp * np.array([1 / n for n in range(1, L + 1)])
    pmax = np.flip(np.maximum.accumulate(np.flip(precision)))
    return np.sum(pmax[ic.nonzero()[0]]) / ntrue


def mAP(predictions, targets, categories, thresholds=(0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95)):
    "Vision.py:1749-1800: mean of mAP1 over categories and thresholds"
    N, C = len(predictions), len(categories)
    targs = [[[] for _ in range(N)] for _ in range(C)]
    preds = [[[] for _ in range(N)] for _ in range(C)]
    scores = [[[] for _ in range(N)] for _ in range(C)]
    for i in range(N):
        pb, pc, cs = predictions[i]
        for j in range(len(pb)):
            preds[pc[j]][i].append(pb[j]); scores[pc[j]][i].append(cs[j])
        for b, c in targets[i]:
            targs[c][i].append(b)
    vals = np.zeros((len(thresholds), C))
    for c in range(C):
        for j, t in enumerate(thresholds):
            vals[j, c] = mAP1(targs[c], preds[c], scores[c], t)
    return np.mean(vals)


def compute_max_overlaps(objects_batch, anchors):
    "Vision.py:1666-1694: (mean over images of mean over objects of max IoU with any anchor, flat list of the maxima)"
    all_max, per_img = [], []
    for objs in objects_batch:
        objs = np.asarray(objs, F32)
        objs = objs[objs >= 0].reshape(-1, 4)
        if len(objs) == 0:
            continue
        m = jaccard(objs, np.asarray(anchors, F32)).max(axis=1)
        all_max += list(m)
        per_img.append(m.mean())
    return (float(np.mean(per_img)) if per_img else 0.0), all_max
