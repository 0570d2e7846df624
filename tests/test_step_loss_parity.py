"""Step-loss parity protocol of SURVEY.md §8(d): same seeded init + same synthetic batches + dropout off -> the per-step loss
of (i) the CPU oracle (restated nets + restated Optimizer.step) and (ii) the 1-GPU HIP path driven by the product Learner
must agree to <= 1e-3 relative over 100 steps.  Heads: collaborative filtering (ML-100K shape), structured data, a small
AWD-LSTM language model (state carried between batches), and a ResNet classifier with BatchNorm in training mode."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import assert_close
from oracle import reference_math as RM
from oracle import reference_nets as RNets
from oracle import reference_text as RT
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'
STEPS = 100


class _Data:
    def __init__(self, batches, bs, target_type, **kw):
        self.train_dl = self.val_dl = batches
        self.bs, self.target_type = bs, target_type
        self.__dict__.update(kw)


def _learner(net, batches, bs, target_type, optimizer, loss_func='default', **data_kw):
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    return Learner('/tmp/nnl_parity', _Data(batches, bs, target_type, **data_kw), net, optimizer=optimizer, loss_func=loss_func)


def _oracle_run(net, loss_of_batch, batches, lrs_of_name, wd, kind, keep_mode=False, **opt_kw):
    if not keep_mode:
        net.train()
    names = [n for n, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]
    state = RM.OptimState(params)
    lrs = [lrs_of_name(n) for n in names]
    losses = []
    for b in batches:
        for p in params:
            p.grad = None
        loss = loss_of_batch(net, b)
        loss.backward()
        losses.append(loss.item())
        RM.optimizer_step(params, [p.grad for p in params], state, lrs, [wd] * len(params), kind, **opt_kw)
    return np.array(losses)


def _to_dev(b):
    f = lambda t: [f(v) for v in t] if isinstance(t, (list, tuple)) else t.to(DEV)
    return f(b[0]), f(b[1])


def test_collab_100_steps():
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterNet
    g = torch.Generator().manual_seed(1234)
    batches = [(torch.stack([torch.randint(0, 943, (64,), generator=g), torch.randint(0, 1682, (64,), generator=g)], 1),
                torch.randint(1, 6, (64,), generator=g).float()) for _ in range(STEPS)]
    torch.manual_seed(0)
    onet = RNets.CollabFilterNet(943, 1682, 30, [0.8, 5.2])
    pnet = CollabFilterNet(943, 1682, 30, [0.8, 5.2])
    pnet.load_state_dict(onet.state_dict())
    ref = _oracle_run(onet, lambda n, b: nn.MSELoss()(n(b[0]), b[1]), batches, lambda n: 1e-2, 1e-4, 'adam')
    learner = _learner(pnet, batches, 64, 'cont', 'Adam')
    learner.init_optimizer(wd=1e-4)
    learner.model.train()
    got = np.array([learner.train1minibatch(*_to_dev(b), 1e-2) for b in batches])
    assert_close(got, ref, 1e-3, 0, 'collab loss curve')
    assert ref[-1] < ref[0]


def test_tabular_100_steps():
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    cards = [116, 5, 4, 13, 53, 13, 4, 8]
    rs = np.random.RandomState(1236)
    batches = []
    for _ in range(STEPS):
        xcat = torch.from_numpy(np.stack([rs.randint(0, c, size=256) for c in cards], 1).astype(np.int64))
        batches.append(([xcat, torch.from_numpy(rs.standard_normal((256, 6)).astype(np.float32))],
                        torch.from_numpy((5 + 7 * rs.rand(256)).astype(np.float32))))
    dims = [RNets.embedding_dim(c) for c in cards]
    onet = RNets.StructuredDataNet('cont', list(zip(cards, dims)), 6, [200, 100, 1], output_range=[5, 12])
    pnet = StructuredDataNet('cont', len(cards), 6, [{i: i for i in range(c)} for c in cards], [200, 100, 1], output_range=[5, 12])
    synth.fill_module_(onet, seed=7); synth.fill_module_(pnet, seed=7)
    group = lambda n: 1 if n.startswith('head') else 0
    ref = _oracle_run(onet, lambda n, b: nn.MSELoss()(n(b[0][0], b[0][1]), b[1]), batches, lambda n: [1e-3, 2e-3][group(n)], 1e-3, 'adam')
    learner = _learner(pnet, batches, 256, 'cont', 'Adam')
    learner.init_optimizer(wd=1e-3)
    learner.model.train()
    got = np.array([learner.train1minibatch(*_to_dev(b), [1e-3, 2e-3]) for b in batches])
    assert_close(got, ref, 1e-3, 0, 'tabular loss curve')


def test_language_model_100_steps_with_state_carry():
    from neuralnetworklibrary_amd.Applications.Text import LanguageModelNet, RegSeqCrossEntropyLoss, _Vocab
    V, bs, bptt = 60, 8, 12
    g = torch.Generator().manual_seed(1237)
    stream = torch.randint(2, V, (bs, STEPS * bptt + 1), generator=g)
    batches = [(stream[:, i * bptt:(i + 1) * bptt].contiguous(), stream[:, i * bptt + 1:(i + 1) * bptt + 1].contiguous()) for i in range(STEPS)]
    onet = RT.LanguageModelNet(V, 1, bs, E=16, Hh=24, L=3)
    stoi = {('tok%d' % i): i for i in range(V)}
    stoi['_pad_'] = 1
    del stoi['tok1']
    vocab = _Vocab(stoi, bs)
    vocab.target_type = 'lang_model'
    pnet = LanguageModelNet(vocab, enc_drops=[0., 0., 0., 0.], dec_drop=0., emb_dim=16, hidden_size=24, num_layers=3)
    for net in (onet, pnet):                              # same names -> same seeded values (as tests/test_text.py)
        synth.fill_module_(net, seed=9)
        with torch.no_grad():
            net.enc.word_embed.embed.weight.mul_(0.3)
    assert [n for n, _ in onet.named_parameters()] == [n for n, _ in pnet.named_parameters()]
    lm_group = lambda n: 0 if '.lstms.' in n else 1
    ref = _oracle_run(onet, lambda n, b: RT.reg_seq_cross_entropy(n(b[0]), b[1], 2.0, 1.0)[0], batches,
                      lambda n: [2e-3, 3e-3][lm_group(n)], 1e-6, 'adam', betas=(0.8, 0.99), clip=0.4)
    learner = _learner(pnet, batches, bs, 'lang_model', 'Adam', loss_func=RegSeqCrossEntropyLoss(2.0, 1.0), stoi=stoi)
    learner.init_optimizer(wd=1e-6, clip=0.4)
    learner.model.train()
    got = np.array([learner.train1minibatch(*_to_dev(b), [2e-3, 3e-3], betas_batch=(0.8, 0.99)) for b in batches])
    assert_close(got, ref, 1e-3, 0, 'LM loss curve')


def test_resnet_classifier_steps_train_mode_bn():
    """ResNet-34 body + default head at 64x64, bs 16, BatchNorm in TRAINING mode, SGD momentum, 16 steps.  At this size the
    last stages normalise over 2x2x16 values per channel and the trajectory is chaotic: the REFERENCE ARITHMETIC ITSELF
    drifts by 1e-4 after one step and 1e-2 after six between fp32 and fp64 (DESIGN.md, G6 note), so a fixed 1e-3 bound is
    meaningless here.  Criterion: the HIP path stays as close to the fp64 oracle as fp32 does (3x the running maximum of the
    fp32-vs-fp64 separation of two fp32 oracle runs — plain, and from weights perturbed by <= 1 ulp — floored at the 95th
    percentile, + 1e-3 relative); step 0 (identical weights) must agree to 1e-4.  The well-conditioned
    frozen-BN step is pinned by golden G6 (tests/test_vision_gpu.py)."""
    from neuralnetworklibrary_amd.Applications import Vision as V
    N, S, steps = 16, 64, 16
    g = torch.Generator().manual_seed(1235)
    batches = [(torch.randn(N, 3, S, S, generator=g), torch.randint(0, 2, (N,), generator=g)) for _ in range(4)]
    seq = [batches[i % 4] for i in range(steps)]
    group = lambda n: 2 if n.startswith('head') else (0 if int(n.split('.')[1]) < 6 else 1)
    lr3 = [2e-4, 5e-4, 1e-3]

    def oracle(dtype, perturb_seed=None):
        net = RNets.ImageClassificationNet(RNets.resnet34(), 2, 512, drops=(0., 0.))
        synth.fill_module_(net, seed=3)
        if perturb_seed is not None:                      # weights moved by <= 1 ulp: the conditioning yardstick
            gen = torch.Generator().manual_seed(perturb_seed)
            with torch.no_grad():
                for p in net.parameters():
                    p.mul_(1.0 + 2.0 ** -23 * (torch.randint(0, 3, p.shape, generator=gen).float() - 1.0))
        net = net.to(dtype)
        data = [(x.to(dtype), y) for x, y in seq]
        return _oracle_run(net, lambda n, b: nn.CrossEntropyLoss()(n(b[0]), b[1]), data, lambda n: lr3[group(n)], 1e-4, 'sgd', momentum=0.9)
    ref32, ref64 = oracle(torch.float32), oracle(torch.float64)
    pert32 = oracle(torch.float32, perturb_seed=1)

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'
    pnet = V.ImageClassificationNet(D, V.models.resnet34(), head=[[512], [0., 0.]])
    synth.fill_module_(pnet, seed=3)
    learner = _learner(pnet, seq, N, 'single_label', 'SGD_Mom', sz=(S, S), categories={0: 'a', 1: 'b'})
    learner.init_optimizer(wd=1e-4)
    learner.model.train()
    got = np.array([learner.train1minibatch(*_to_dev(b), lr3) for b in seq])
    assert_close(got[:1], ref64[:1], 1e-4, 0, 'step 0')
    # the separation fp32 has ALREADY shown from fp64 up to step i (two fp32 samples: plain, and from eps-perturbed weights)
    gap = np.maximum.accumulate(np.maximum(np.abs(ref32 - ref64), np.abs(pert32 - ref64)))
    bound = 3 * np.maximum(gap, np.quantile(gap, 0.95)) + 1e-3 * np.abs(ref64) + 1e-5
    err = np.abs(got - ref64)
    assert (err <= bound).all(), 'step %d: |hip - fp64| = %.3e, fp32-vs-fp64 gap %.3e' % (int(np.argmax(err - bound)), err.max(), gap.max())
