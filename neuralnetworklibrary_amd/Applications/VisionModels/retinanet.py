"""ResNet blocks, FPN, RetinaNet heads and anchors of the drop-in API (mirror of the reference's
Applications/VisionModels/retinanet.py:26-495; the yhenon/pytorch-retinanet lineage is credited there).

Same classes, constructor arguments, sub-module names (=> state_dict keys) and outputs.  Differences:
  * every convolution is `HipConv2d`, an nn.Conv2d whose forward/backward run the fp32-MFMA implicit-GEMM
    kernels (ops.conv2d, K1); conv+ReLU pairs of the heads use the fused ReLU epilogue;
  * activations flow in NHWC (channels_last) between layers;
  * `AnchorGenerator` caches the anchors per image shape ON THE DEVICE instead of rebuilding them in numpy and
    copying H2D on every forward (retinanet.py:485-495);
  * nms / BBoxPredictor (inference post-processing, retinanet.py:500-812): threshold + decode + clip, the top_k sort and
    the greedy same-class NMS run in HIP kernels (csrc/detect.hip) for the whole batch at once; the list filters on the
    few survivors (relative thresholds, inclusions, cross-class duplicates, max_boxes) are host logic as upstream.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import ops
from ...General.Core import default_device

__all__ = ['HipConv2d', 'conv3x3', 'BasicBlock', 'Bottleneck', 'PyramidFeatures', 'RegressionModel',
           'ClassificationModel', 'RetinaNet', 'retinanet18', 'retinanet34', 'retinanet50', 'retinanet101',
           'retinanet152', 'get_anchor_set', 'get_anchor_shifts', 'AnchorGenerator', 'intersections', 'jaccard', 'nms',
           'BBoxPredictor']


class HipConv2d(nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict) computed by the HIP implicit-GEMM kernels.  The weight keeps its logical
    [K,C,R,S] shape but is STORED channels_last (= KRSC, the kernels' filter layout), so no per-step re-layout is needed
    and the weight gradient comes back in the parameter's own layout; state_dict load / save are unaffected."""
    fuse_relu = False
    nnl_hip_conv = True               # ops.prepare_backward batches the filter transposes of these modules

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x, grad_slot=None, give_slot=None):
        if self.dilation != (1, 1) or self.groups != 1 or self.stride[0] != self.stride[1] \
                or self.padding[0] != self.padding[1] or self.padding_mode != 'zeros':
            raise NotImplementedError('HipConv2d: only the symmetric, dense convolutions the reference uses')
        # (inside ops.shared_params — the detection heads on the five pyramid levels — weight / bias are this call's aliases)
        return ops.conv2d(x, ops.fan_param(self.weight), ops.fan_param(self.bias), self.stride[0], self.padding[0], relu=self.fuse_relu,
                          grad_slot=grad_slot, give_slot=give_slot)


class HipMaxPool2d(nn.MaxPool2d):
    "nn.MaxPool2d (floor mode, no dilation) on the NHWC HIP kernels of csrc/pool.hip (same tie rule as torch)"

    def forward(self, x):
        k, s, p = self.kernel_size, self.stride, self.padding
        if self.dilation != 1 or self.ceil_mode or self.return_indices or not all(isinstance(v, int) for v in (k, s, p)):
            raise NotImplementedError('HipMaxPool2d: only the square floor-mode pooling the reference uses')
        return ops.maxpool2d(x, k, s, p)


class _ConvReLU(HipConv2d):
    "conv followed by nn.ReLU in the reference (heads, retinanet.py:192-193 etc.): ReLU fused in the epilogue"
    fuse_relu = True


def conv3x3(in_planes, out_planes, stride=1):
    "3x3 convolution with padding, no bias (retinanet.py:26-28)"
    return HipConv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    "conv3x3-BN-ReLU-conv3x3-BN (+downsample(x)) -ReLU   (retinanet.py:30-59)"
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # the shortcut's gradient w.r.t. x is added inside conv1's dgrad kernel (ops.GradSlot) instead of by autograd: the
        # closing BN parks it there for an identity shortcut, the downsample convolution for a projection shortcut
        slot = ops.GradSlot() if (x.requires_grad and torch.is_grad_enabled()) else None
        out = ops.conv_bn_act(self.conv1, self.bn1, x, relu=True, conv_slot=slot)
        if self.downsample is None:
            return ops.conv_bn_act(self.conv2, self.bn2, out, residual=x, relu=True, bn_slot=slot)
        return ops.conv_bn_act(self.conv2, self.bn2, out, residual=_shortcut(self.downsample, x, slot), relu=True)


class Bottleneck(nn.Module):
    "1x1-BN-ReLU-3x3(stride)-BN-ReLU-1x1(x4)-BN (+downsample(x)) -ReLU   (retinanet.py:61-97)"
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = HipConv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = HipConv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = HipConv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        slot = ops.GradSlot() if (x.requires_grad and torch.is_grad_enabled()) else None
        out = ops.conv_bn_act(self.conv1, self.bn1, x, relu=True, conv_slot=slot)
        out = ops.conv_bn_act(self.conv2, self.bn2, out, relu=True)
        if self.downsample is None:
            return ops.conv_bn_act(self.conv3, self.bn3, out, residual=x, relu=True, bn_slot=slot)
        return ops.conv_bn_act(self.conv3, self.bn3, out, residual=_shortcut(self.downsample, x, slot), relu=True)


class _Downsample(nn.Sequential):
    "Sequential(conv1x1(stride), BN) (retinanet.py:344-348) evaluated with the fused BN epilogue"

    def forward(self, x, give_slot=None):
        return ops.conv_bn_act(self[0], self[1], x, relu=False, conv_give=give_slot)


def _shortcut(downsample, x, slot):
    "projection shortcut; a _Downsample hands its input gradient to the block's first conv through `slot`"
    return downsample(x, give_slot=slot) if isinstance(downsample, _Downsample) else downsample(x)


class PyramidFeatures(nn.Module):
    "FPN P3..P7 from C3, C4, C5 (retinanet.py:101-148)"

    def __init__(self, C3_size, C4_size, C5_size, feature_size=256):
        super().__init__()
        self.P5_1 = HipConv2d(C5_size, feature_size, kernel_size=1, stride=1, padding=0)
        self.P5_upsampled = nn.Upsample(scale_factor=2, mode='nearest')
        self.P5_2 = HipConv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P4_1 = HipConv2d(C4_size, feature_size, kernel_size=1, stride=1, padding=0)
        self.P4_upsampled = nn.Upsample(scale_factor=2, mode='nearest')
        self.P4_2 = HipConv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P3_1 = HipConv2d(C3_size, feature_size, kernel_size=1, stride=1, padding=0)
        self.P3_2 = HipConv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P6 = HipConv2d(C5_size, feature_size, kernel_size=3, stride=2, padding=1)
        self.P7_1 = nn.ReLU()
        self.P7_2 = HipConv2d(feature_size, feature_size, kernel_size=3, stride=2, padding=1)

    @staticmethod
    def _fusable(lateral, big, small):
        "1x1 / stride 1 lateral conv over `big` whose output is exactly twice `small` in both directions, C % 16 == 0"
        return (big.is_cuda and lateral.kernel_size == (1, 1) and lateral.stride == (1, 1) and big.shape[1] % 16 == 0
                and big.shape[2] == 2 * small.shape[2] and big.shape[3] == 2 * small.shape[3])

    def forward(self, inputs):
        C3, C4, C5 = inputs
        P5_x = self.P5_1(C5)
        if self._fusable(self.P4_1, C4, P5_x) and self._fusable(self.P3_1, C3, C4):
            # `P5_upsampled + P4_1(C4)` / `P3_1(C3) + P4_upsampled` (retinanet.py:131-141): upsample + add in the lateral
            # convolution's epilogue (the upsampled maps are never written)
            P4_x = ops.conv_add_upsampled(C4, self.P4_1.weight, self.P4_1.bias, P5_x)
            P5_x = self.P5_2(P5_x)
            P3_x = self.P3_2(ops.conv_add_upsampled(C3, self.P3_1.weight, self.P3_1.bias, P4_x))
            P4_x = self.P4_2(P4_x)
        else:
            P5_up = self.P5_upsampled(P5_x)
            P5_x = self.P5_2(P5_x)
            P4_x = P5_up + self.P4_1(C4)
            P4_up = self.P4_upsampled(P4_x)
            P4_x = self.P4_2(P4_x)
            P3_x = self.P3_2(self.P3_1(C3) + P4_up)
        P6_x = self.P6(C5)
        P7_x = self.P7_2(self.P7_1(P6_x))
        return [P3_x, P4_x, P5_x, P6_x, P7_x]


class _Head(nn.Module):
    """Shared structure of RegressionModel / ClassificationModel: [bn0][drop0] -> 4 x (conv3x3 -> ReLU -> [bn]
    -> [drop]) -> output conv (retinanet.py:163-185, 232-258)."""

    def _build(self, num_features_in, feature_size, n_out, bn, drop):
        self.drop0 = nn.Dropout(drop[0]) if drop else None
        self.drop = nn.Dropout(drop[1]) if drop else None
        self.bn0 = nn.BatchNorm2d(num_features_in, momentum=0.01) if bn else None
        chans = [num_features_in, feature_size, feature_size, feature_size, feature_size]
        for i in range(1, 5):
            setattr(self, 'conv%d' % i, _ConvReLU(chans[i - 1], feature_size, kernel_size=3, padding=1))
            setattr(self, 'act%d' % i, nn.ReLU())          # kept for module-tree parity; fused into the conv
            setattr(self, 'bn%d' % i, nn.BatchNorm2d(feature_size, momentum=0.01) if bn else None)
        self.output = HipConv2d(feature_size, n_out, kernel_size=3, padding=1)

    def _trunk(self, x, sigmoid=False):
        if self.bn0:
            x = self.bn0(x)
        if self.drop0:
            x = self.drop0(x)
        out = x
        for i in range(1, 5):
            out = getattr(self, 'conv%d' % i)(out)            # conv + ReLU (fused)
            bn = getattr(self, 'bn%d' % i)
            if bn:
                out = bn(out)
            if self.drop:
                out = self.drop(out)
        o = self.output
        if sigmoid and out.shape[1] % 16 == 0 and o.stride == (1, 1) and o.padding[0] == o.padding[1]:
            # `output_act` (nn.Sigmoid, retinanet.py:286) in the epilogue of the output convolution; its backward gate and the bias
            # gradient are one pass (nnl_act_gate_colsum)
            return ops.conv2d(out, o.weight, o.bias, 1, o.padding[0], relu=2)
        out = o(out)
        return self.output_act(out) if sigmoid else out


class RegressionModel(_Head):
    "bs x C x H x W -> bs x (H*W*A) x 4, cell-major / anchor-minor (retinanet.py:150-217)"

    def __init__(self, num_features_in, num_anchors=9, feature_size=256, bn=False, drop=None):
        super().__init__()
        self._build(num_features_in, feature_size, num_anchors * 4, bn, drop)

    def forward(self, x):
        out = self._trunk(x)                                  # logical [bs, A*4, H, W], physical NHWC
        return out.permute(0, 2, 3, 1).reshape(out.shape[0], -1, 4)


class ClassificationModel(_Head):
    "bs x C x H x W -> bs x (H*W*A) x K probabilities (sigmoid inside, retinanet.py:219-295)"

    def __init__(self, num_features_in, num_anchors=9, num_classes=80, feature_size=256, bn=False, drop=None):
        super().__init__()
        self.num_classes, self.num_anchors = num_classes, num_anchors
        self._build(num_features_in, feature_size, num_anchors * num_classes, bn, drop)
        self.output_act = nn.Sigmoid()

    def forward(self, x):
        out = self._trunk(x, sigmoid=True)
        return out.permute(0, 2, 3, 1).reshape(out.shape[0], -1, self.num_classes)


class RetinaNet(nn.Module):
    "ResNet backbone + FPN + heads (retinanet.py:299-386); init: conv N(0, sqrt(2/(k*k*out))), BN (1,0), prior .01"

    def __init__(self, num_classes, block, layers):
        self.inplanes = 64
        super().__init__()
        self.conv1 = HipConv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = HipMaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        last = 'conv2' if block == BasicBlock else 'conv3'
        fpn_sizes = [getattr(self.layer2[layers[1] - 1], last).out_channels,
                     getattr(self.layer3[layers[2] - 1], last).out_channels,
                     getattr(self.layer4[layers[3] - 1], last).out_channels]
        self.fpn = PyramidFeatures(fpn_sizes[0], fpn_sizes[1], fpn_sizes[2])
        self.regressionModel = RegressionModel(256)
        self.classificationModel = ClassificationModel(256, num_classes=num_classes)
        self.AnchorGenerator = AnchorGenerator()
        self.BBoxPredictor = BBoxPredictor()                      # retinanet.py:325
        init_retina_modules(self.modules())
        prior = 0.01
        nn.init.constant_(self.classificationModel.output.weight, 0)
        nn.init.constant_(self.classificationModel.output.bias, -np.log((1.0 - prior) / prior))
        nn.init.constant_(self.regressionModel.output.weight, 0)
        nn.init.constant_(self.regressionModel.output.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = _Downsample(
                HipConv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def stem(self, x):
        return ops.conv_bn_relu_maxpool(self.conv1, self.bn1, self.maxpool, x)

    def forward(self, img_batch):
        x = self.stem(img_batch)
        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x4 = self.layer4(x3)
        features = self.fpn([x2, x3, x4])
        with ops.shared_params([self.regressionModel, self.classificationModel], len(features)):     # one gradient sum per shared parameter
            reg = torch.cat([self.regressionModel(f) for f in features], dim=1)
            clas = torch.cat([self.classificationModel(f) for f in features], dim=1)
        return [self.AnchorGenerator(img_batch), reg, clas]


def init_retina_modules(modules):
    "conv ~ N(0, sqrt(2/(kh*kw*out))), BN weight 1 / bias 0 (retinanet.py:327-333; Vision.py:1425-1431)"
    for m in modules:
        if isinstance(m, nn.Conv2d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            nn.init.normal_(m.weight, mean=0.0, std=np.sqrt(2 / n))
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def retinanet18(num_classes, pretrained=False, **kwargs):
    return RetinaNet(num_classes, BasicBlock, [2, 2, 2, 2], **kwargs)


def retinanet34(num_classes, pretrained=False, **kwargs):
    return RetinaNet(num_classes, BasicBlock, [3, 4, 6, 3], **kwargs)


def retinanet50(num_classes, pretrained=False, **kwargs):
    return RetinaNet(num_classes, Bottleneck, [3, 4, 6, 3], **kwargs)


def retinanet101(num_classes, pretrained=False, **kwargs):
    return RetinaNet(num_classes, Bottleneck, [3, 4, 23, 3], **kwargs)


def retinanet152(num_classes, pretrained=False, **kwargs):
    return RetinaNet(num_classes, Bottleneck, [3, 8, 36, 3], **kwargs)


def retinanet(weights_path=None):
    """ResNet-50 RetinaNet with the COCO-pretrained weights when a real checkpoint file is given
    (retinanet.py:430-435; the repository ships only an LFS pointer, so the default is random init)."""
    model = RetinaNet(80, Bottleneck, [3, 4, 6, 3])
    if weights_path is not None:
        model.load_state_dict(torch.load(weights_path, map_location='cpu'))
    return model


# ---- anchors (retinanet.py:439-495) -----------------------------------------------------------------------

def get_anchor_set(ratios=[0.5, 1, 2], scales=[2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)]):
    "Base anchors around (0,0): area = scale, width/height = ratio; ratio-major, scale-minor order (:439-451)"
    S = np.tile(scales, len(ratios))
    Rt = np.repeat(ratios, len(scales))
    Hs, Ws = S / np.sqrt(Rt), S * np.sqrt(Rt)
    return np.array([-Ws / 2, -Hs / 2, Ws / 2, Hs / 2]).T


def get_anchor_shifts(shape, stride, anchors):
    "Place the A base anchors at every cell centre (i+0.5)*stride of an (H,W) grid; cell-major order (:453-471)"
    sx = (np.arange(0, shape[1]) + 0.5) * stride
    sy = (np.arange(0, shape[0]) + 0.5) * stride
    sx, sy = np.meshgrid(sx, sy)
    shifts = np.stack([sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()], axis=1)      # [K,4]
    return (shifts[:, None, :] + anchors[None, :, :]).reshape(-1, 4)


class AnchorGenerator(object):
    "Anchors for pyramid levels 3..7 (stride 2^l, base size 2^(l+2)); cached per (H, W, device) (:473-495)"

    def __init__(self, ratios=[0.5, 1, 2], scales=[2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)]):
        self.pyramid_levels = [3, 4, 5, 6, 7]
        self.strides = [2 ** x for x in self.pyramid_levels]
        self.sizes = [2 ** (x + 2) for x in self.pyramid_levels]
        self.ratios, self.scales = np.array(ratios), np.array(scales)
        self.anchor_set = get_anchor_set(ratios, scales)
        self._cache = {}

    def numpy_anchors(self, H, W):
        shape = np.array([H, W])
        out = []
        for lvl, stride, size in zip(self.pyramid_levels, self.strides, self.sizes):
            grid = (shape + 2 ** lvl - 1) // (2 ** lvl)
            out.append(get_anchor_shifts(grid, stride, size * self.anchor_set))
        return np.concatenate(out)

    def __call__(self, img_batch):
        key = (int(img_batch.shape[2]), int(img_batch.shape[3]), str(img_batch.device))
        if key not in self._cache:
            a = torch.as_tensor(self.numpy_anchors(key[0], key[1]), dtype=torch.float32)
            self._cache[key] = a.to(img_batch.device)
        return self._cache[key]


# ---- BBoxPredictor and associated functions (reference retinanet.py:498-812) -------------------------------------------------
def intersections(Boxes1, Boxes2):
    "[n,m] intersection areas of min-max boxes given as numpy arrays (retinanet.py:500-509)"
    B1, B2 = np.expand_dims(Boxes1, axis=1), np.expand_dims(Boxes2, axis=0)
    inter_w = (np.minimum(B1[:, :, 2], B2[:, :, 2]) - np.maximum(B1[:, :, 0], B2[:, :, 0])).clip(0, None)
    inter_h = (np.minimum(B1[:, :, 3], B2[:, :, 3]) - np.maximum(B1[:, :, 1], B2[:, :, 1])).clip(0, None)
    return inter_w * inter_h


def jaccard(Boxes1, Boxes2):
    "[n,m] jaccard index (IoU) of min-max boxes given as numpy arrays (retinanet.py:511-521)"
    areas1 = (Boxes1[:, 2] - Boxes1[:, 0]) * (Boxes1[:, 3] - Boxes1[:, 1])
    areas2 = (Boxes2[:, 2] - Boxes2[:, 0]) * (Boxes2[:, 3] - Boxes2[:, 1])
    inter = intersections(Boxes1, Boxes2)
    return inter / (np.expand_dims(areas1, axis=1) + np.expand_dims(areas2, axis=0) - inter)


def _device_nms(cand, bs, cap, top_k, max_overlap, device):
    """sorted top_k + greedy same-class NMS for a batch of candidate lists on the GPU (nnl_nms); returns per image the
    survivors as numpy arrays (boxes [m,4] f32, classes [m] i64, scores [m] f32) — ONE device->host copy for the batch."""
    from ..._lib import check, lib, ptr, stream
    cbox, ccls, cscore, corder, ccount = cand
    top_k = int(min(top_k, cap))
    kbox = torch.empty(bs, top_k, 4, dtype=torch.float32, device=device)
    kcls = torch.empty(bs, top_k, dtype=torch.int32, device=device)
    kscore = torch.empty(bs, top_k, dtype=torch.float32, device=device)
    kcount = torch.empty(bs, dtype=torch.int32, device=device)
    wsb = int(lib.nnl_nms_workspace_bytes(bs, top_k))
    ws = torch.empty(wsb // 4 + 1, dtype=torch.float32, device=device)
    check(lib.nnl_nms(ptr(cbox), ptr(ccls), ptr(cscore), ptr(corder), ptr(ccount), bs, cap, top_k, float(max_overlap), ptr(kbox),
                      ptr(kcls), ptr(kscore), ptr(kcount), ptr(ws), wsb, stream()))
    counts = kcount.cpu().numpy()
    mx = int(counts.max()) if bs else 0
    hb, hc, hs = kbox[:, :mx].cpu().numpy(), kcls[:, :mx].cpu().numpy().astype(np.int64), kscore[:, :mx].cpu().numpy()
    return [(hb[i, :counts[i]], hc[i, :counts[i]], hs[i, :counts[i]]) for i in range(bs)]


def _drop(seq, idxs):
    idxs = set(int(i) for i in idxs)
    return [v for i, v in enumerate(seq) if i not in idxs]


def _prune(pred_boxes, pred_classes, conf_scores, rel_thresh, max_boxes, dup, inc):
    """The list filters that follow the suppression loop in the reference's nms (retinanet.py:613-705), on the survivors
    (lists sorted by descending score): relative thresholds, single inclusions of one class, cross-class duplicates,
    max_boxes."""
    S, C, B = list(conf_scores), list(pred_classes), list(pred_boxes)
    if rel_thresh:
        t1, t2 = rel_thresh
        for i in range(len(S)):
            if S[i] < t1 * S[0]:
                S, C, B = S[:i], C[:i], B[:i]
                break
        if len(S) > 1:
            sv, cv = np.array(S), np.array(C)
            # j is dropped when an earlier i of its class has S[j] < t2*S[i]; with descending scores the binding i is the
            # FIRST member of the class
            first = {}
            kill = []
            for j in range(len(S)):
                c = int(cv[j])
                if c not in first:
                    first[c] = j
                elif sv[j] < t2 * sv[first[c]]:
                    kill.append(j)
            S, C, B = _drop(S, kill), _drop(C, kill), _drop(B, kill)
    if inc and len(C):
        thr, inc_classes = inc
        L = len(C)
        pc, pb = np.array(C), np.array(B)
        eq = (pc[:, None] == pc[None, :]).astype(int)
        area = (pb[:, 2] - pb[:, 0]) * (pb[:, 3] - pb[:, 1])
        ratios = intersections(pb, pb) / area
        ratios2 = area[None, :] / area[:, None]
        big = ((ratios * eq > thr).astype(int) - np.identity(L, int)) * (ratios2 > 0.25).astype(int)
        single = [int(i) for i in (big.sum(axis=1) == 1).nonzero()[0] if int(C[i]) not in inc_classes]
        partners = {int(np.argmax(big[i])) for i in single}
        kill = []
        for i in set(single) - partners:
            j = int(np.argmax(big[i]))
            if S[i] < 0.75 * S[j]:
                kill.append(i)
            elif S[j] < 0.75 * S[i]:
                kill.append(j)
        S, C, B = _drop(S, kill), _drop(C, kill), _drop(B, kill)
    if dup:
        thr, pairs = dup
        pairs = set(tuple(p) for p in pairs)
        again = True
        while again and len(B):
            again = False
            jac = jaccard(np.array(B), np.array(B))
            L = len(B)
            for i in range(L - 1):
                hit = next((j for j in range(i + 1, L)
                            if jac[i, j] > thr and (C[i], C[j]) in pairs and S[j] < 0.75 * S[i]), -1)
                if hit >= 0:
                    del S[hit], C[hit], B[hit]
                    again = True
                    break
    return B[:max_boxes], C[:max_boxes], S[:max_boxes]


def nms(pred_boxes, pred_classes, conf_scores, max_overlap=0.5, rel_thresh=None, top_k=1000, max_boxes=20, dup=None,
        inc=None, print_it=False):
    """Non-maximum suppression + box pruning for ONE image (reference retinanet.py:523-711; same arguments and outputs:
    lists of np.array([xmin,ymin,xmax,ymax]), classes and scores in descending score order).  pred_boxes [N,4],
    pred_classes [N], conf_scores [N] are CUDA tensors: the sort and the suppression loop run in csrc/detect.hip."""
    if len(pred_boxes) == 0:
        return [], [], []
    from ..._lib import require_cuda
    require_cuda(pred_boxes, pred_classes, conf_scores)
    dev, N = pred_boxes.device, len(pred_boxes)
    cand = (pred_boxes.detach().float().contiguous().view(1, N, 4), pred_classes.detach().to(torch.int32).contiguous().view(1, N),
            conf_scores.detach().float().contiguous().view(1, N), None, torch.tensor([N], dtype=torch.int32, device=dev))
    b, c, s = _device_nms(cand, 1, N, top_k, max_overlap, dev)[0]
    if print_it:
        print('after non-max-supress'); print(len(b), len(c), len(s))
    B, C, S = _prune(list(b), list(c), list(s), rel_thresh, max_boxes, dup, inc)
    if print_it:
        print('after pruning / max_boxes'); print(len(B), len(C), len(S)); print('')
    return B, C, S


class BBoxPredictor(object):
    """Predicted bounding boxes from RetinaNet activations, pruned by thresholding and NMS (reference retinanet.py:713-812;
    same arguments, `mean` / `std` semantics and outputs).  The whole batch is decoded, sorted and suppressed on the GPU with
    four kernel launches and one device->host copy of the survivors."""

    def __init__(self, mean=[0., 0., 0., 0.], std=[0.1, 0.1, 0.2, 0.2]):
        self.mean, self.std = np.asarray(mean, np.float32), np.asarray(std, np.float32)

    def __call__(self, img_batch, reg, clas, anchors, thresh=0.05, max_overlap=0.5, rel_thresh=None, top_k=1000,
                 max_boxes=20, dup=None, inc=None):
        from ..._lib import check, lib, ptr, require_cuda, stream
        require_cuda(reg, clas, anchors)
        bs, _, height, width = img_batch.shape
        reg, clas, anchors = reg.detach().float().contiguous(), clas.detach().float().contiguous(), anchors.detach().float().contiguous()
        A, K, dev = anchors.shape[0], clas.shape[2], reg.device
        cbox = torch.empty(bs, A, 4, dtype=torch.float32, device=dev)
        ccls = torch.empty(bs, A, dtype=torch.int32, device=dev)
        cscore = torch.empty(bs, A, dtype=torch.float32, device=dev)
        corder = torch.empty(bs, A, dtype=torch.int32, device=dev)
        ccount = torch.empty(bs, dtype=torch.int32, device=dev)
        check(lib.nnl_bbox_decode(ptr(anchors), ptr(reg), ptr(clas), bs, A, K, self.mean.ctypes.data, self.std.ctypes.data,
                                  float(thresh), float(width), float(height), ptr(cbox), ptr(ccls), ptr(cscore), ptr(corder),
                                  ptr(ccount), stream()))
        PredBoxes, PredClasses, ConfScores = [], [], []
        for b, c, s in _device_nms((cbox, ccls, cscore, corder, ccount), bs, A, top_k, max_overlap, dev):
            B, C, S = _prune(list(b), list(c), list(s), rel_thresh, max_boxes, dup, inc) if len(b) else ([], [], [])
            PredBoxes.append(B); PredClasses.append(C); ConfScores.append(S)
        return PredBoxes, PredClasses, ConfScores
