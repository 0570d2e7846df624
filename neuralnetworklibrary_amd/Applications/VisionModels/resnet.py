"""torchvision-compatible ResNet (same topology, module names and state_dict keys as
torchvision.models.resnet*, which the reference passes as `arch` to ImageClassificationNet — Planet.ipynb cell 17,
consumed at Applications/Vision.py:1211-1212,1225-1228) built from the HIP-backed blocks of retinanet.py.
torchvision itself is not a dependency; if it is installed its ResNet is accepted too (Vision.default_cut)."""
import torch.nn as nn

from .retinanet import BasicBlock, Bottleneck, HipConv2d, HipMaxPool2d, _Downsample
from ... import ops

__all__ = ['ResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152']


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = HipConv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = HipMaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():                      # torchvision's init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

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

    def forward(self, x):
        x = ops.conv_bn_relu_maxpool(self.conv1, self.bn1, self.maxpool, x)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x).flatten(1)
        return ops.linear(x, self.fc.weight, self.fc.bias, relu=False)


class ResNetBody(nn.Sequential):
    """`nn.Sequential(*list(resnet.children())[:-2])` (Vision.py:1212) with the stem conv->BN->ReLU evaluated
    through the fused BN epilogue; children / indices / state_dict keys are those of the plain Sequential."""

    def forward(self, x):
        mods = list(self.children())
        if len(mods) >= 4 and isinstance(mods[0], HipConv2d) and isinstance(mods[1], nn.BatchNorm2d) \
                and isinstance(mods[2], nn.ReLU) and isinstance(mods[3], nn.MaxPool2d):
            x = ops.conv_bn_relu_maxpool(mods[0], mods[1], mods[3], x)          # BN + ReLU + pooling in one pass
            mods = mods[4:]
        elif len(mods) >= 3 and isinstance(mods[0], HipConv2d) and isinstance(mods[1], nn.BatchNorm2d) \
                and isinstance(mods[2], nn.ReLU):
            x = ops.conv_bn_act(mods[0], mods[1], x, relu=True)
            mods = mods[3:]
        for m in mods:
            x = m(x)
        return x


def resnet18(pretrained=False, **kw):
    return ResNet(BasicBlock, [2, 2, 2, 2], **kw)


def resnet34(pretrained=False, **kw):
    return ResNet(BasicBlock, [3, 4, 6, 3], **kw)


def resnet50(pretrained=False, **kw):
    return ResNet(Bottleneck, [3, 4, 6, 3], **kw)


def resnet101(pretrained=False, **kw):
    return ResNet(Bottleneck, [3, 4, 23, 3], **kw)


def resnet152(pretrained=False, **kw):
    return ResNet(Bottleneck, [3, 8, 36, 3], **kw)
