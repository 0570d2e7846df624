"""End-to-end drop-in flows on the GPU through the public API only: data object -> Learner -> fit / evaluate / predict, with
the MI355X opt-ins (device-resident loaders, whole-step hipGraph replay) switched on.  The loss must go down and the
graph / eager runs must agree on what they learn."""
import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _setup():
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False


def _ratings(n=4000, n_user=60, n_item=40, seed=0):
    rs = np.random.RandomState(seed)
    pu, pi = rs.standard_normal((n_user, 3)), rs.standard_normal((n_item, 3))
    u, i = rs.randint(0, n_user, n), rs.randint(0, n_item, n)
    r = np.clip(3 + (pu[u] * pi[i]).sum(1) + 0.1 * rs.standard_normal(n), 1, 5).astype('float32')
    return pd.DataFrame({'user': u, 'item': i, 'rating': r})


@pytest.mark.parametrize('graphs', [False, True])
def test_collab_fit_device_resident(graphs):
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterDataObj, CollabFilterNet
    from neuralnetworklibrary_amd.General.Learner import Learner
    _setup()
    df = _ratings()
    labels = [{u: k for k, u in enumerate(sorted(df.user.unique()))}, {m: k for k, m in enumerate(sorted(df.item.unique()))}]
    data = CollabFilterDataObj(df[:3200], df[3200:], 'user', 'item', 'rating', labels, bs=256, device_resident=True, seed=1)
    torch.manual_seed(0)
    net = CollabFilterNet(len(labels[0]), len(labels[1]), 8, [0.8, 5.2])
    learner = Learner('/tmp/nnl_e2e', data, net, optimizer='Adam')
    if graphs:
        learner.use_graphs(True)
    before = learner.evaluate('val')[0]
    learner.fit(5e-2, 6, wd=1e-5)
    after = learner.evaluate('val')[0]
    assert after < 0.5 * before, (before, after)
    pred = learner.predict('val')
    assert pred.shape == (800,) and np.isfinite(pred).all() and 0.8 <= pred.min() and pred.max() <= 5.2
    if graphs:
        assert sum(g.graph is not None for g in learner._graphs.values()) >= 1


def test_tabular_fit_device_resident_with_graphs():
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet, StructuredDataObj, StructuredDataset
    from neuralnetworklibrary_amd.General.Learner import Learner
    _setup()
    rs = np.random.RandomState(2)
    N, cards = 3000, [7, 5, 11]
    xcat = np.stack([rs.randint(0, c, N) for c in cards], 1)
    xcont = rs.standard_normal((N, 4)).astype('float32')
    eff = [rs.standard_normal(c) for c in cards]
    y = (sum(e[xcat[:, j]] for j, e in enumerate(eff)) + xcont[:, 0] - 0.5 * xcont[:, 1] + 0.05 * rs.standard_normal(N)).astype('float32')
    tr = StructuredDataset(pd.DataFrame(xcat[:2400]), pd.DataFrame(xcont[:2400]), y[:2400], 'cont')
    va = StructuredDataset(pd.DataFrame(xcat[2400:]), pd.DataFrame(xcont[2400:]), y[2400:], 'cont')
    labels = [{k: k for k in range(c)} for c in cards]
    data = StructuredDataObj(tr, va, labels, None, bs=200, device_resident=True, seed=3)
    torch.manual_seed(0)
    net = StructuredDataNet('cont', 3, 4, labels, [64, 32, 1], dropout_levels=(0.02, 0.02, [0, 0.1, 0.05]))
    learner = Learner('/tmp/nnl_e2e', data, net, optimizer='Adam').use_graphs(True)
    before = learner.evaluate('val')[0]
    learner.fit_one_cycle([3e-3, 3e-3], 8, wd=1e-4)
    after = learner.evaluate('val')[0]
    assert after < 0.25 * before, (before, after)
    assert len(learner.loss_sched) == 8 * len(data.train_dl)


def test_image_classifier_workflow_freeze_unfreeze_findlr_save_load():
    """The fast.ai-style workflow of the reference notebooks on the GPU path: train the head with the body frozen, find_lr,
    unfreeze with per-layer-group learning rates, bn_freeze, checkpoint round trip with optimizer state, predict / evaluate."""
    from neuralnetworklibrary_amd.Applications import Vision as V
    _setup()
    N, S = 8, 64
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(4 * N, 3, S, S, generator=g)
    ys = (xs.mean(dim=(1, 2, 3)) > 0).long()                       # a learnable rule
    batches = [(xs[i * N:(i + 1) * N].to(DEV), ys[i * N:(i + 1) * N].to(DEV)) for i in range(4)]

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'neg', 1: 'pos'}, N, 'single_label'
        train_dl, val_dl = batches, batches[:2]
    torch.manual_seed(1)
    net = V.ImageClassificationNet(D, V.models.resnet18(), head=[[32], [0.1, 0.1]])
    learner = V.ImageLearner('/tmp/nnl_e2e_img', D, net, optimizer='SGD_Mom')
    learner.freeze()
    assert all(not p.requires_grad for p in net.body.parameters()) and all(p.requires_grad for p in net.head.parameters())
    body0 = [p.detach().clone() for p in net.body.parameters()]
    learner.fit(1e-2, 2, wd=1e-4)
    assert all(torch.equal(a, b) for a, b in zip(body0, net.body.parameters())), 'frozen body must not move'
    w_before = [p.detach().clone() for p in net.parameters()]
    learner.find_lr(lr_min=1e-5, lr_max=1e-1, plot=False)
    assert all(torch.equal(a, b) for a, b in zip(w_before, net.parameters())), 'find_lr must restore the weights'
    learner.unfreeze()
    learner.fit_one_cycle([1e-4, 3e-4, 3e-3], 2, wd=1e-4)
    assert not all(torch.equal(a, b) for a, b in zip(body0, net.body.parameters()))
    learner.bn_freeze('all')
    learner.fit([1e-4, 3e-4, 1e-3], 1)
    learner.bn_unfreeze()
    learner.save('ckpt', save_optimizer=True)
    before = learner.evaluate('val')
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05)
    assert learner.evaluate('val')[0] != before[0]
    learner.load('ckpt', saved_optimizer=True)
    after = learner.evaluate('val')
    assert after[0] == before[0] and after[1] == before[1]
    probs, labels = learner.predict('val')
    assert probs.shape == (2 * N, 2) and np.allclose(probs.sum(1), 1, atol=1e-5) and set(labels.tolist()) <= {0, 1}
    learner.fit(1e-3, 1)                                            # training continues after a load


def test_detection_workflow_fit_predict_map():
    "RetinaNet: a few SSD-loss training steps, then Learner.predict('val') through the GPU BBoxPredictor and compute_mAP."
    from neuralnetworklibrary_amd.Applications import Vision as V
    _setup()
    S, K = 128, 3
    g = torch.Generator().manual_seed(0)

    def target(n):
        boxes = -torch.ones(n, 2, 4); cats = -torch.ones(n, 2, dtype=torch.long)
        for i in range(n):
            x0, y0 = torch.randint(0, 60, (2,), generator=g).tolist()
            w, h = torch.randint(30, 60, (2,), generator=g).tolist()
            boxes[i, 0] = torch.tensor([x0, y0, x0 + w, y0 + h], dtype=torch.float32)
            cats[i, 0] = int(torch.randint(0, K, (1,), generator=g))
        return boxes, cats
    train = []
    for _ in range(2):
        b, c = target(2)
        train.append((torch.randn(2, 3, S, S, generator=g).to(DEV), [b.to(DEV), c.to(DEV)]))
    val = []
    val_targets = []
    for cats2 in ([0, 1], [2, 0]):                                       # every category has ground truth (else mAP1 is 0/0, as upstream)
        b = torch.tensor([[[10., 12., 60., 70.], [50., 40., 100., 90.]]])
        c = torch.tensor([cats2])
        val.append((torch.randn(1, 3, S, S, generator=g).to(DEV), [b.to(DEV), c.to(DEV)]))
        val_targets.append([(b[0, j].numpy(), int(c[0, j])) for j in range(2)])

    class VDS:
        images = [{'scale': 1.0}, {'scale': 0.5}]
        y = val_targets

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b', 2: 'c'}, 2, 'bbox'
        train_dl, val_dl, val_ds = train, val, VDS
    torch.manual_seed(0)
    net = V.ObjectDetectionNet(K)
    learner = V.ImageLearner('/tmp/nnl_e2e_det', D, net, optimizer='SGD_Mom', loss_func=V.SSD_loss(0.5, 0.25, 2.0))
    learner.fit(1e-3, 2, wd=1e-4)
    assert len(learner.loss_sched) == 4 and all(np.isfinite(learner.loss_sched))
    preds = learner.predict('val', thresh=0.0, max_boxes=5)              # thresh 0: an untrained net still returns boxes
    assert len(preds) == 2
    for (boxes, classes, scores), img in zip(preds, VDS.images):
        assert 0 < len(boxes) <= 5 and len(boxes) == len(classes) == len(scores)
        assert all(scores[i] >= scores[i + 1] for i in range(len(scores) - 1))
        assert all(0 <= int(c) < K for c in classes)
        assert np.all(np.array(boxes) <= S / img['scale'] + 1e-3)         # rescaled to the original image
    m = learner.compute_mAP(predictions=preds, mAP_thresholds=[0.5])
    assert 0.0 <= m <= 1.0


def test_language_model_workflow_device_resident_loader():
    "AWD-LSTM LM: device-resident LanguageModelDataObj -> fit (hidden state carried across batches) -> evaluate with accuracy"
    from neuralnetworklibrary_amd.Applications import Text as TX
    _setup()
    V_ = 40
    rs = np.random.RandomState(0)
    stoi = {('t%d' % i): i for i in range(V_)}
    stoi['_pad_'] = 1
    del stoi['t1']

    def ds(n_texts):
        class DS:
            pass
        d = DS()
        # a learnable pattern: token i is followed by (i + 3) % V
        d.texts = []
        for _ in range(n_texts):
            start = int(rs.randint(2, V_))
            d.texts.append([(start + 3 * k) % (V_ - 2) + 2 for k in range(60)])
        d.num_tokens = sum(len(t) for t in d.texts)
        d.stoi = stoi
        return d
    data = TX.LanguageModelDataObj(ds(40), ds(8), None, bs=8, bptt=24, device_resident=True)   # (bptt//2 - 9 must stay > 0: Text.py:270-276)
    x0, y0 = next(iter(data.val_dl))
    assert x0.is_cuda and torch.equal(x0[:, 1:], y0[:, :-1])
    torch.manual_seed(0)
    net = TX.LanguageModelNet(data, enc_drops=[0.02, 0.05, 0.05, 0.02], dec_drop=0.02, emb_dim=24, hidden_size=32, num_layers=2)
    from neuralnetworklibrary_amd.General.Learner import Learner
    learner = Learner('/tmp/nnl_e2e_lm', data, net, optimizer='Adam', loss_func=TX.RegSeqCrossEntropyLoss(2.0, 1.0))
    before = learner.evaluate('val', metrics=[TX.LanguageModelAccuracy()])
    learner.fit([3e-3, 3e-3], 3, wd=1e-6, clip=0.4, betas=(0.8, 0.99))
    after = learner.evaluate('val', metrics=[TX.LanguageModelAccuracy()])
    assert after[0] < before[0] and after[1][0] > before[1][0], (before, after)
