"""Vision application of the drop-in API: model + loss side of the reference's Applications/Vision.py
(§5 ImageClassificationNet :1203-1373, §6 ObjectDetectionNet and the SSD loss :1376-1663).

On the hot path and HIP-backed: every convolution (ops.conv2d, K1), the classifier head linears, and the
detection loss — `SSD_loss.__call__` is ONE fused anchor-match + focal + smooth-L1 kernel per batch
(ops.retina_loss, K6) with no host synchronisation, replacing the reference's per-image Python loop
(Vision.py:1636), per-positive-anchor scalar indexing loop (:1593) and `.nonzero()` syncs (:1506-1507).
Out of scope here (CPU image decode / augmentation / display / mAP evaluation): the
cv2/skimage-based Transform / ImageDataset / ImageDataObj classes and ImageLearner's visualisation helpers.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..General.Core import *          # noqa: F401,F403
from ..General.Layers import *        # noqa: F401,F403
from ..General.Learner import *       # noqa: F401,F403
from ..General.LossesMetrics import * # noqa: F401,F403
from ..General.Optimizer import *     # noqa: F401,F403
from ..General.Core import TEN, separate_bn_layers
from ..General.Layers import AdaptiveConcatPool2d, Flatten, FullyConnectedNet
from ..General.Learner import Learner
from .VisionModels import vmods
from .VisionModels import resnet as models      # stands in for `torchvision.models` (Vision.py:8)
from .VisionModels.resnet import ResNetBody
from .. import ops

try:  # a torchvision ResNet (if installed) is accepted by default_cut / default_split as well
    import torchvision.models as _tvm
    _RESNET_TYPES = (models.ResNet, _tvm.ResNet)
except Exception:  # torchvision is not installed in this image
    _RESNET_TYPES = (models.ResNet,)

imagenet_stats = [np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])]
alternate_stats = [np.array([0.5, 0.5, 0.5]), np.array([0.5, 0.5, 0.5])]
Pascal_thresholds = [0.5]
COCO_thresholds = [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]


def jaccard(Boxes1, Boxes2):
    """IoU of every pair (b1, b2): [n,4] x [m,4] -> [n,m], min-max boxes, no +1 (Vision.py:234-256)."""
    if (len(Boxes1) == 0) or (len(Boxes2) == 0):
        return TEN([])
    B1, B2 = Boxes1.float(), Boxes2.float()
    a1 = (B1[:, 2] - B1[:, 0]) * (B1[:, 3] - B1[:, 1])
    a2 = (B2[:, 2] - B2[:, 0]) * (B2[:, 3] - B2[:, 1])
    b1, b2 = B1.unsqueeze(1), B2.unsqueeze(0)
    iw = (torch.min(b1[:, :, 2], b2[:, :, 2]) - torch.max(b1[:, :, 0], b2[:, :, 0])).clamp(min=0)
    ih = (torch.min(b1[:, :, 3], b2[:, :, 3]) - torch.max(b1[:, :, 1], b2[:, :, 1])).clamp(min=0)
    inter = iw * ih
    return inter / (a1.unsqueeze(1) + a2.unsqueeze(0) - inter)


# ---- §5 image classification ---------------------------------------------------------------------------------

def default_cut(model):
    """Cut a known body arch before its pooling / classifier (Vision.py:1205-1219)."""
    if isinstance(model, _RESNET_TYPES):
        return ResNetBody(*list(model.children())[:-2])
    if isinstance(model, (vmods.ResNeXt101_32x4d, vmods.ResNeXt101_64x4d, vmods.InceptionV4)):
        return model.features
    if isinstance(model, vmods.SENet):
        return nn.Sequential(*list(model.children())[:5])
    return model


def default_split(precut_body, body):
    """Split a known body arch into 2 layer groups about half way (Vision.py:1221-1242)."""
    kids = list(body.children())
    if isinstance(precut_body, _RESNET_TYPES) or isinstance(precut_body, (vmods.ResNeXt101_32x4d, vmods.ResNeXt101_64x4d)):
        return [nn.Sequential(*kids[:6]), nn.Sequential(*kids[6:])]
    if isinstance(precut_body, vmods.SENet):
        return [nn.Sequential(*kids[:3]), nn.Sequential(*kids[3:])]
    if isinstance(precut_body, vmods.InceptionV4):
        return [nn.Sequential(*kids[:11]), nn.Sequential(*kids[11:])]
    return [body]


def _num_features(body, data):
    """Channel count of the body's output.  The reference pushes a zero image through the training-mode body
    (Vision.py:1312-1313), which as a side effect moves every BN running stat one momentum step towards (0, 0); with a
    GPU present the same probe runs here (on the device: the HIP body has no CPU path) so the buffers match.  Without
    a GPU (host-logic tests) the count is read off the last conv / batch-norm."""
    if torch.cuda.is_available():
        dev = default_device()
        with torch.no_grad():
            return body.to(dev)(torch.zeros(1, 3, data.sz[0], data.sz[1], device=dev)).shape[1]
    last = None
    for m in body.modules():
        if isinstance(m, nn.BatchNorm2d):
            last = m.num_features
        elif isinstance(m, nn.Conv2d):
            last = m.out_channels
    if last is not None:
        return last
    dev = default_device()
    with torch.no_grad():
        return body.to(dev)(torch.zeros(1, 3, data.sz[0], data.sz[1], device=dev)).shape[1]


class ImageClassificationNet(nn.Module):
    """Pretrained-style `body` + `head` classifier (Vision.py:1244-1337).  head default:
    AdaptiveConcatPool2d -> Flatten -> FullyConnectedNet([2*nfeats, 512, ncats], drops [.25,.25]);
    layer_groups = body groups (default_split) + [head]."""

    def __init__(self, data, arch, head='default', cutpoint='default', splits='default'):
        super().__init__()
        if cutpoint is None:
            self.body = arch
        elif cutpoint == 'default':
            self.body = default_cut(arch)
        elif type(cutpoint) == int:
            self.body = nn.Sequential(*list(arch.children())[:cutpoint])

        if isinstance(head, nn.Module):
            self.head = head
        else:
            if type(head) == list:
                layer_sizes, drops = head[0], head[1]
            elif head == 'default':
                layer_sizes, drops = [512], [0.25, 0.25]
            nfeats = _num_features(self.body, data)
            ncats = len(data.categories)
            fully_connected = FullyConnectedNet([2 * nfeats] + layer_sizes + [ncats], drops)
            self.head = nn.Sequential(AdaptiveConcatPool2d(), Flatten(), fully_connected)

        if splits is None:
            body_groups = [self.body]
        elif type(splits) == str:
            body_groups = default_split(arch, self.body)
        elif type(splits) == nn.ModuleList:
            body_groups = [G for G in splits]
        elif type(splits) == list:
            layers = list(self.body.children())
            idxs = [0] + splits + [len(layers)]
            body_groups = [nn.Sequential(*layers[idxs[i]:idxs[i + 1]]) for i in range(len(idxs) - 1)]

        self.layer_groups = body_groups + [self.head]
        self.param_groups = separate_bn_layers(self.layer_groups)

    def forward(self, x_batch):
        return self.head(self.body(x_batch))


class ImageClassificationEnsembleNet(nn.Module):
    "Weighted average of softmax / sigmoid outputs of several classifiers (Vision.py:1339-1373)."

    def __init__(self, models, weights=None, correction='single_label'):
        super().__init__()
        n = len(models)
        self.weights = weights if weights else [1 / n] * n
        self.correction = correction
        self.models = nn.ModuleList(models)
        self.layer_groups = models
        self.param_groups = separate_bn_layers(self.layer_groups)

    def forward(self, x):
        if self.correction == 'single_label':
            return sum(w * F.log_softmax(m(x), dim=1).exp() for w, m in zip(self.weights, self.models))
        if self.correction == 'multi_label':
            return sum(w * m(x).sigmoid() for w, m in zip(self.weights, self.models))


# ---- §6 object detection ------------------------------------------------------------------------------------------

class ObjectDetectionNet(nn.Module):
    """RetinaNet (ResNet-50 + FPN) with re-initialised classifier / regressor heads (Vision.py:1382-1471).
    The reference loads COCO weights from an LFS blob that is not in the repository (retinanet.py:430-435); pass
    `pretrained_path` to load a real checkpoint, otherwise the backbone keeps its seeded random init."""

    def __init__(self, num_classes, ratios=[0.5, 1, 2], scales=[2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)],
                 prior=0.01, feature_size=256, bn=False, drop=None, pretrained_path=None):
        super().__init__()
        R = vmods.retinanet
        model = R.retinanet(pretrained_path)
        self.layer0 = ResNetBody(model.conv1, model.bn1, model.relu, model.maxpool)
        self.layer1, self.layer2, self.layer3, self.layer4 = model.layer1, model.layer2, model.layer3, model.layer4
        self.resnet = nn.ModuleList([self.layer0, self.layer1, self.layer2, self.layer3, self.layer4])
        self.fpn = model.fpn

        num_anchors = len(ratios) * len(scales)
        self.classifier = R.ClassificationModel(256, num_anchors, num_classes, feature_size, bn, drop)
        self.regressor = R.RegressionModel(256, num_anchors, feature_size, bn, drop)
        self.head = nn.ModuleList([self.classifier, self.regressor])
        R.init_retina_modules(self.head.modules())
        nn.init.constant_(self.classifier.output.weight, 0)
        nn.init.constant_(self.classifier.output.bias, -np.log((1.0 - prior) / prior))
        nn.init.constant_(self.regressor.output.weight, 0)
        nn.init.constant_(self.regressor.output.bias, 0)

        self.layer_groups = [self.resnet, self.fpn, self.head]
        self.param_groups = separate_bn_layers(self.layer_groups)
        self.AnchorGenerator = R.AnchorGenerator(ratios, scales)
        self.BBoxPredictor = R.BBoxPredictor()                    # Vision.py:1444

    def forward(self, x):
        """x [bs,3,H,W] -> [anchors [N,4], reg [bs,N,4], clas [bs,N,num_classes]]"""
        x0 = self.layer0(x)
        x1 = self.layer1(x0)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x4 = self.layer4(x3)
        features = self.fpn([x2, x3, x4])
        # the head parameters are shared by the five levels: one gradient-sum launch per tensor instead of four autograd adds
        with ops.shared_params([self.regressor, self.classifier], len(features)):
            reg = torch.cat([self.regressor(f) for f in features], dim=1)
            clas = torch.cat([self.classifier(f) for f in features], dim=1)
        return [self.AnchorGenerator(x), reg, clas]


def match_anchors_objects(objects, anchors, pos_thresh=0.5, neg_thresh=0.4):
    """Per-image anchor/object matching (Vision.py:1474-1511): returns pos_idxs, neg_idxs, matches.  Helper API
    (torch ops); the training path uses the fused kernel in SSD_loss."""
    dev = anchors.device
    if len(objects) == 0:
        return (torch.zeros(0, dtype=torch.long, device=dev), torch.arange(len(anchors), device=dev),
                -torch.ones(len(anchors), dtype=torch.long, device=dev))
    max_values, max_idxs = torch.max(jaccard(objects, anchors), dim=0)
    pos = max_values > pos_thresh
    matches = pos.long() * (max_idxs + 1) - 1
    return pos.nonzero().view(-1), (max_values < neg_thresh).nonzero().view(-1), matches


def focal_loss_retina(pred, target, alpha=0.25, gamma=2.0):
    "Focal loss of one image, summed and divided by max(#positives, 1) (Vision.py:1513-1530); helper API."
    p = pred.clamp(1e-4, 1.0 - 1e-4)
    pt = p * target + (1 - p) * (1 - target)
    w = (alpha * target + (1 - alpha) * (1 - target)) * (1 - pt).pow(gamma)
    losses = -w * (target * torch.log(p) + (1 - target) * torch.log(1 - p))
    return losses.sum() / target.sum().clamp(min=1)


def smoothL1_loss_retina(anchs, pred_shift, target):
    "Smooth-L1 (beta = 1/9) on encoded box deltas, mean over n_pos*4 (Vision.py:1532-1566); helper API."
    aw, ah = anchs[:, 2] - anchs[:, 0], anchs[:, 3] - anchs[:, 1]
    ax, ay = anchs[:, 0] + 0.5 * aw, anchs[:, 1] + 0.5 * ah
    tw, th = target[:, 2] - target[:, 0], target[:, 3] - target[:, 1]
    tx, ty = target[:, 0] + 0.5 * tw, target[:, 1] + 0.5 * th
    tw, th = tw.clamp(min=1), th.clamp(min=1)
    true_shift = torch.stack(((tx - ax) / aw, (ty - ay) / ah, torch.log(tw / aw), torch.log(th / ah))).t()
    true_shift = true_shift / torch.tensor([[0.1, 0.1, 0.2, 0.2]], device=anchs.device)
    diff = torch.abs(true_shift - pred_shift)
    losses = 0.5 * 9 * diff.pow(2) * (diff < 1 / 9).float() + (diff - 0.5 / 9) * (diff >= 1 / 9).float()
    return losses.mean()


def ssd1(anchors, bboxes, cats, reg, clas, alpha=0.25, gamma=2.0):
    "(reg_loss, clas_loss) of ONE image (Vision.py:1568-1605) through the fused kernel."
    M = max(len(bboxes), 1)
    B = -torch.ones(1, M, 4, device=reg.device)
    Cc = -torch.ones(1, M, dtype=torch.long, device=reg.device)
    if len(bboxes):
        B[0, :len(bboxes)] = bboxes
        Cc[0, :len(cats)] = cats
    out = ops.retina_loss(anchors, reg.unsqueeze(0), clas.unsqueeze(0), B, Cc, 0.5, alpha, gamma)
    return out[1], out[2]


class SSD_loss(object):
    """(1-beta)*smoothL1 + beta*focal, batch mean of per-image losses (Vision.py:1607-1644).  One fused kernel per
    batch (fwd) + one (bwd); `.reg_loss` / `.clas_loss` are stashed as 0-dim device tensors as in the reference."""

    def __init__(self, beta=0.5, alpha=0.25, gamma=2.0):
        self.beta, self.alpha, self.gamma = beta, alpha, gamma

    def __call__(self, activ, target):
        BBoxes, Cats = target[0], target[1]
        anchors, reg, clas = activ[0], activ[1], activ[2]
        out = ops.retina_loss(anchors, reg, clas, BBoxes, Cats, self.beta, self.alpha, self.gamma)
        self.reg_loss, self.clas_loss = out[1].detach(), out[2].detach()
        return out[0]


class SSD_RegLoss(object):
    "Metric exposing SSD_loss.reg_loss (Vision.py:1646-1654)."
    def __init__(self, loss_func):
        self.loss_func = loss_func

    def __call__(self, pred, target):
        return self.loss_func.reg_loss


class SSD_ClasLoss(object):
    "Metric exposing SSD_loss.clas_loss (Vision.py:1656-1663)."
    def __init__(self, loss_func):
        self.loss_func = loss_func

    def __call__(self, pred, target):
        return self.loss_func.clas_loss


# ---- §6.3 other detection metrics (Vision.py:1666-1800) -------------------------------------------------------------------
class ComputeMaxOverlaps(object):
    """Mean over images of the mean over ground-truth objects of the maximum jaccard overlap with any anchor box — how well
    the anchors cover the objects (Vision.py:1666-1694).  Also accumulates every maximum in self.max_overlaps."""

    def __init__(self):
        self.max_overlaps = []

    def __call__(self, activ, target):
        Objects, anchors, bs = target[0], activ[0], len(target[0])
        batch_means = []
        for i in range(bs):
            objects = Objects[i][Objects[i] >= 0].view(-1, 4)
            if len(objects) == 0:
                continue
            mx = jaccard(objects, anchors).max(dim=1)[0].detach().cpu().numpy()
            self.max_overlaps += list(mx)
            batch_means.append(mx.mean())
        return TEN(np.array(batch_means).mean() if batch_means else 0.0)


def mAP1(targs, preds, scores, thresh):
    """Average precision of one category at one jaccard threshold over a dataset (Vision.py:1696-1747): each ground-truth box
    marks its best-overlapping prediction correct if the overlap exceeds `thresh`; the area under the max-smoothed precision
    curve, divided by the number of ground-truth boxes."""
    from .VisionModels.retinanet import jaccard as np_jaccard
    is_correct, all_scores = [], []
    for t, p, s in zip(targs, preds, scores):
        ok = [0] * len(p)
        if len(p) > 0 and len(t) > 0:
            jac = np_jaccard(np.array(t, dtype=np.float32), np.array(p, dtype=np.float32))
            best = jac.argmax(axis=1)
            for j, k in enumerate(best):
                if jac[j, k] > thresh:
                    ok[int(k)] = 1
        is_correct += ok
        all_scores += list(s)
    ic = np.array([c for _, c in sorted(zip(all_scores, is_correct), reverse=True)])
    ntrue = sum(len(t) for t in targs)
    tp = np.cumsum(ic)
    precision = tp * np.array([1 / n for n in range(1, len(ic) + 1)])
    smoothed = np.flip(np.maximum.accumulate(np.flip(precision)))
    return np.sum(smoothed[ic.nonzero()[0]]) / ntrue


def mAP(predictions, targets, categories, thresholds=COCO_thresholds, verbose=True):
    """Mean average precision over categories and jaccard thresholds (Vision.py:1749-1800).  predictions[i] =
    [pred_boxes, pred_classes, conf_scores] as returned by learner.predict('val'); targets[i] = [(box, category), ...]."""
    N, C = len(predictions), len(categories)
    targs = [[[] for _ in range(N)] for _ in range(C)]
    preds = [[[] for _ in range(N)] for _ in range(C)]
    scores = [[[] for _ in range(N)] for _ in range(C)]
    for i in range(N):
        pred_boxes, pred_classes, conf_scores = predictions[i]
        for j in range(len(pred_boxes)):
            c = pred_classes[j]
            preds[c][i].append(pred_boxes[j])
            scores[c][i].append(conf_scores[j])
        for b, c in targets[i]:
            targs[c][i].append(b)
    vals = np.zeros((len(thresholds), C))
    for c in range(C):
        for j, thresh in enumerate(thresholds):
            vals[j, c] = mAP1(targs[c], preds[c], scores[c], thresh)
            if verbose:
                print('cat =', c, ':', categories[c], ' thresh =', thresh)
                print('cat-thresh mAP = ', vals[j, c])
                print('')
    if verbose:
        print('Overall mAP = ', np.mean(vals))
    return np.mean(vals)


class ImageLearner(Learner):
    """Learner for image data (Vision.py:1803-1812): inherits fit / evaluate / predict unchanged, plus compute_mAP for
    object detection.  The visualisation, TTA and pycocotools conveniences of the reference's ImageLearner are UI / external
    tooling (out of scope)."""

    def compute_mAP(self, predictions=None, thresh=0.05, max_overlap=0.5, rel_thresh=None, top_k=1000, max_boxes=20,
                    dup=None, inc=None, mAP_thresholds=COCO_thresholds):
        "mAP of the validation set, target_type 'bbox' only (Vision.py:2123-2140)"
        categories, targets = self.data.categories, self.data.val_ds.y
        if predictions is None:
            predictions = self.predict('val', True, thresh, max_overlap, rel_thresh, top_k, max_boxes, dup, inc)
        return mAP(predictions, targets, categories, mAP_thresholds)
