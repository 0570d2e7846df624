"""ORACLE — TEST INFRASTRUCTURE ONLY.  Deterministic synthetic parameters / inputs shared by the golden generator
(reference side) and the tests (oracle + product side), so that big models need no multi-MB weight fixtures:
both sides fill the SAME closed-form values into identically named tensors."""
import zlib

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


def fill_reference_init_(module, seed=0):
    """In-place deterministic init with the DISTRIBUTIONS of the reference's own constructors, in closed form (numpy legacy
    RandomState, so the reference side in oracle/gen_golden*.py and the oracle / product side in tests/ hold bit-identical values
    without a multi-MB weight fixture):
      * 4-d weights (convolutions): N(0, 2 / (k_h k_w out_channels)) — `RetinaNet.__init__`, reference retinanet.py:327-330
        (what torchvision's ResNet does as kaiming_normal_(mode='fan_out'));
      * 2-d weights (linear layers): N(0, 2 / fan_in) — `initialize_modules(.., nn.init.kaiming_normal_)`, reference
        General/Layers.py:137, General/Core.py:159-175; their biases 0 (Core.py:171);
      * BatchNorm scale 1 and shift 0 (retinanet.py:331-333; torch's default for the head's BatchNorm1d);
      * every other 1-d parameter (conv biases): 0."""
    with torch.no_grad():
        for j, (name, p) in enumerate(module.named_parameters()):
            tag = 104729 * (seed + 1) + j
            if p.dim() == 4:
                p.copy_(torch.from_numpy(synth_array(p.shape, tag, (2.0 / (p.shape[0] * p.shape[2] * p.shape[3])) ** 0.5)))
            elif p.dim() >= 2:
                p.copy_(torch.from_numpy(synth_array(p.shape, tag, (2.0 / int(np.prod(p.shape[1:]))) ** 0.5)))
            elif name.endswith('weight'):
                p.fill_(1.0)
            else:
                p.zero_()
    return module


def curve_labels(n, tag, ncls=2):
    return torch.from_numpy(np.random.RandomState(200003 + int(tag)).randint(0, ncls, size=(n,)).astype(np.int64))


def fill_lm_reference_init_(net, seed=0):
    """The reference LanguageModelNet's own init distributions in closed form: embedding ~ U(-0.1, 0.1) with the pad row 0
    (reference Text.py:461-462), every LSTM weight / bias ~ U(-1/sqrt(H), 1/sqrt(H)) with H the layer's hidden size (torch's
    nn.LSTM default, which `WeightDropLSTM1` keeps, Text.py:483-488); the decoder weight is tied to the embedding."""
    with torch.no_grad():
        seen = set()
        for j, (name, p) in enumerate(net.named_parameters()):
            if id(p) in seen:
                continue
            seen.add(id(p))
            # keyed by the parameter's NAME (the reference lists weight_hh_l0_raw last, other builds may not)
            tag = (zlib.crc32(name.encode()) + 130003 * (seed + 1)) % (2 ** 32)
            u = np.random.RandomState(tag).random_sample(tuple(p.shape)).astype(np.float32) * 2.0 - 1.0
            if 'embed' in name:
                p.copy_(torch.from_numpy(u * 0.1))
                p[1].zero_()                                      # pad token 1
            else:
                H = p.shape[0] // 4
                p.copy_(torch.from_numpy(u * (1.0 / H ** 0.5)))
    return net


def lm_stream(V, bs, length, tag, n_sub=512, p_follow=0.75):
    """LEARNABLE token stream [bs, length] for the language-model loss-curve fixture: tokens come from a fixed subset of `n_sub`
    vocabulary entries (ids >= 4, spread over the whole vocabulary) and follow a first-order chain — with probability `p_follow`
    the next token is a fixed function of the current one, else uniform over the subset.  (A uniform stream over V = 47 343
    tokens, SURVEY.md §8d's benchmark input, has nothing to learn: its loss curve stays at ln V and would not notice a broken
    update.)  numpy legacy RandomState: bit-stable."""
    rs = np.random.RandomState(300007 + int(tag))
    sub = 4 + rs.permutation(V - 4)[:n_sub].astype(np.int64)
    idx = rs.randint(0, n_sub, size=(bs, length))
    follow = rs.random_sample((bs, length)) < p_follow
    for t in range(1, length):
        idx[:, t] = np.where(follow[:, t], (3 * idx[:, t - 1] + 7) % n_sub, idx[:, t])
    return sub[idx]


def detection_targets(N, M, S, K, tag):
    """SURVEY.md §8d config 5 targets in closed form: per image m ~ U{1..M} boxes, x0,y0 ~ U(0, 0.58 S), w,h ~ U(0.06 S, 0.41 S)
    (clipped to the image), classes ~ U{0..K-1}; padded with -1 to [N,M,4] / [N,M]."""
    rs = np.random.RandomState(400009 + int(tag))
    boxes = -np.ones((N, M, 4), np.float32)
    cats = -np.ones((N, M), np.int64)
    for i in range(N):
        m = int(rs.randint(1, M + 1))
        xy = rs.uniform(0, 0.58 * S, size=(m, 2))
        wh = rs.uniform(0.06 * S, 0.41 * S, size=(m, 2))
        b = np.concatenate([xy, np.minimum(xy + wh, S - 1.0)], 1)
        boxes[i, :m] = np.round(b).astype(np.float32)
        cats[i, :m] = rs.randint(0, K, size=m)
    return boxes, cats


def curve_batch_images(N, S, tag, amp=0.5):
    """One LEARNABLE synthetic image minibatch for the loss-curve fixtures: 0.5 N(0,1) noise, labels ~ U{0,1}, and the label
    written into channel 0 as a +-amp offset — so a 20-step curve actually descends (pure noise with random labels gives a
    flat, maximally ill-conditioned one)."""
    x, y = synth_input((N, 3, S, S), tag), curve_labels(N, tag)
    x = x * 0.5
    x[:, 0] += amp * (2.0 * y.float() - 1.0).view(-1, 1, 1)
    return x, y


def tame_residual_branches_(net, gamma=0.3):
    """For fixtures that run BatchNorm on its (initial: mean 0, variance 1) running statistics: with no renormalisation the 16 residual
    sums of ResNet-34 double the activation variance per block (logits of several hundred, a saturated loss of ~250 at the reference
    init).  Sets the LAST BatchNorm scale of every residual block (`bn2`, BasicBlock: retinanet.py:43-59) to `gamma`, so that each block
    adds gamma^2 of its branch variance (x4 over the network instead of x65 536).  Works on the reference's, the oracle's and the
    product's modules alike (same attribute names)."""
    import torch
    n = 0
    with torch.no_grad():
        for m in net.modules():
            if hasattr(m, 'conv2') and hasattr(m, 'bn2') and not hasattr(m, 'conv3'):
                m.bn2.weight.fill_(gamma)
                n += 1
    assert n == 16, n
    return net


ROSSMANN_CARDS = [1116, 5, 4, 13, 53, 13, 4, 8, 32, 23, 27, 24, 28, 9, 5, 5] + [10] * 16      # SURVEY.md 8d config 3 (as bench.py)


def rossmann_batch(bs, n_cont, tag):
    """one LEARNABLE Rossmann-shaped minibatch (fixture G16; the tests regenerate it from the tag): categorical columns ~ U{0..c-1},
    continuous ~ N(0,1), target = 8.5 + a fixed linear function of three continuous columns and of the parity of two categorical
    ones + noise, inside the output range [5, 12]"""
    rs = np.random.RandomState(160000 + tag)
    xcat = np.stack([rs.randint(0, c, size=bs) for c in ROSSMANN_CARDS], 1).astype(np.int64)
    xcont = rs.standard_normal((bs, n_cont)).astype(np.float32)
    y = 8.5 + 0.8 * xcont[:, 0] - 0.5 * xcont[:, 3] + 0.3 * xcont[:, 7] + 0.6 * (xcat[:, 1] % 2) - 0.4 * (xcat[:, 5] % 2) + 0.1 * rs.standard_normal(bs)
    return xcat, xcont, np.clip(y, 5.2, 11.8).astype(np.float32)
