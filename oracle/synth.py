"""ORACLE — TEST INFRASTRUCTURE ONLY.  Deterministic synthetic parameters / inputs shared by the golden generator
(reference side) and the tests (oracle + product side), so that big models need no multi-MB weight fixtures:
both sides fill the SAME closed-form values into identically named tensors."""
import numpy as np
import torch


def synth_array(shape, tag, scale=1.0):
    """float32 standard-normal array * scale from numpy's legacy Mersenne-Twister RandomState(tag) — bit-stable across
    machines and numpy versions (legacy generator), unlike torch.manual_seed streams across torch versions."""
    return (np.random.RandomState(int(tag)).standard_normal(tuple(shape)) * scale).astype(np.float32)


def fill_module_(module, seed=0):
    """In-place deterministic init of every parameter (named_parameters order): conv/linear weights ~ N(0, 2/fan_in),
    1-d '*.weight' (BN scale) ~ 1 + 0.1 N(0,1), biases ~ 0.1 N(0,1); buffers are left alone."""
    with torch.no_grad():
        for j, (name, p) in enumerate(module.named_parameters()):
            tag = 7919 * (seed + 1) + j
            if p.dim() >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                a = synth_array(p.shape, tag, (2.0 / fan_in) ** 0.5)
            elif name.endswith('weight'):
                a = 1.0 + synth_array(p.shape, tag, 0.1)
            else:
                a = synth_array(p.shape, tag, 0.1)
            p.copy_(torch.from_numpy(a))
    return module


def synth_input(shape, tag, scale=1.0):
    return torch.from_numpy(synth_array(shape, 100003 + tag, scale))


def fill_detection_net_(net, seed=0):
    """fill_module_ + the conditioning an ObjectDetectionNet needs to keep every activation O(1) WITHOUT relying on batch
    statistics (eval-mode BatchNorm is an affine map): the closing BatchNorm of each Bottleneck (`*.bn3.weight`) is scaled by
    0.2 so 16 residual blocks do not double the signal each; the two head output convolutions get small weights and the
    classifier's output bias sits at -2 (sigmoid outputs ~0.1, unsaturated).  Same closed-form values on the reference side
    (oracle/gen_golden.py g12) and on the oracle / product side (tests)."""
    fill_module_(net, seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith('bn3.weight'):
                p.mul_(0.2)
            elif name in ('classifier.output.weight', 'regressor.output.weight'):
                p.mul_(0.25)
            elif name == 'classifier.output.bias':
                p.sub_(2.0)
    return net
