"""Device-resident input pipelines (device_data.DeviceBatches, the device_resident options of the data objects): host logic,
runs on the CPU device here (same code path with device='cuda')."""
import numpy as np
import pandas as pd
import torch


def test_device_batches_cover_the_dataset_once_per_epoch_and_reshuffle():
    from neuralnetworklibrary_amd.device_data import DeviceBatches
    N = 103
    x = [torch.arange(N).view(N, 1).repeat(1, 3), torch.arange(N).float().view(N, 1)]
    y = torch.arange(N).float()
    dl = DeviceBatches(x, y, bs=16, shuffle=True, device='cpu', seed=5)
    assert len(dl) == 7
    epochs = []
    for _ in range(2):
        seen = []
        for (xc, xf), yb in dl:
            assert xc.shape[1:] == (3,) and torch.equal(xc[:, 0].float(), yb) and torch.equal(xf[:, 0], yb)
            seen.append(yb)
        assert [len(s) for s in seen] == [16] * 6 + [7]
        allv = torch.cat(seen)
        assert torch.equal(allv.sort().values, y)
        epochs.append(allv)
    assert not torch.equal(epochs[0], epochs[1])              # a new permutation every epoch
    again = torch.cat([yb for _, yb in DeviceBatches(x, y, 16, True, 'cpu', seed=5)])
    assert torch.equal(again, epochs[0])                      # deterministic given the seed


def test_rank_slices_partition_each_global_batch():
    from neuralnetworklibrary_amd.device_data import DeviceBatches
    from neuralnetworklibrary_amd.dist import ShardedBatches
    N, bs, world = 50, 8, 2
    x, y = torch.arange(N).view(N, 1), torch.arange(N).float()
    single = list(DeviceBatches(x, y, bs * world, True, 'cpu', seed=9))
    ranks = [list(DeviceBatches(x, y, bs, True, 'cpu', seed=9, rank=r, world=world)) for r in range(world)]
    assert len(single) == len(ranks[0]) == len(ranks[1]) == 4
    for b, (xs, ys) in enumerate(single):
        assert torch.equal(torch.cat([ranks[0][b][1], ranks[1][b][1]]), ys)
        # the same cut dist.ShardedBatches makes of a host-side global batch
        for r in range(world):
            cut = list(ShardedBatches([(xs, ys)], r, world))[0]
            assert torch.equal(cut[1], ranks[r][b][1])


def test_data_objects_device_resident_match_the_dataloader_path():
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterDataObj
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataObj, StructuredDataset
    from neuralnetworklibrary_amd.General.Core import set_default_device
    set_default_device('cpu')
    rs = np.random.RandomState(0)
    df = pd.DataFrame({'u': rs.randint(0, 9, 40), 'm': rs.randint(0, 7, 40), 'r': rs.randint(1, 6, 40).astype('float32')})
    labels = [{u: i for i, u in enumerate(sorted(df.u.unique()))}, {m: i for i, m in enumerate(sorted(df.m.unique()))}]
    host = CollabFilterDataObj(df[:30], df[30:], 'u', 'm', 'r', labels, bs=8, num_workers=0)
    dev = CollabFilterDataObj(df[:30], df[30:], 'u', 'm', 'r', labels, bs=8, num_workers=0, device_resident=True)
    for (xh, yh), (xd, yd) in zip(host.val_dl, dev.val_dl):     # unshuffled loaders yield identical batches
        assert torch.equal(xh, xd) and torch.equal(yh, yd) and xd.dtype == torch.int64 and yd.dtype == torch.float32
    assert len(dev.train_dl) == len(host.train_dl) == 4

    xcat = pd.DataFrame(rs.randint(0, 5, (25, 3))); xcont = pd.DataFrame(rs.standard_normal((25, 2)).astype('float32'))
    yv = rs.rand(25).astype('float32')
    tr, va = StructuredDataset(xcat[:20], xcont[:20], yv[:20], 'cont'), StructuredDataset(xcat[20:], xcont[20:], yv[20:], 'cont')
    hs = StructuredDataObj(tr, va, None, None, bs=6, num_workers=0)
    ds = StructuredDataObj(tr, va, None, None, bs=6, num_workers=0, device_resident=True)
    for ((ch, fh), yh), ((cd, fd), yd) in zip(hs.val_dl, ds.val_dl):
        assert torch.equal(ch, cd) and torch.equal(fh, fd) and torch.equal(yh, yd)
    (c0, f0), y0 = next(iter(ds.train_dl))
    assert c0.shape == (6, 3) and f0.shape == (6, 2) and y0.shape == (6,)


def test_language_model_loader_device_resident_yields_the_same_windows():
    from neuralnetworklibrary_amd.Applications.Text import LanguageModelDataLoader
    from neuralnetworklibrary_amd.General.Core import set_default_device
    set_default_device('cpu')

    class DS:
        texts = [list(range(i, i + 37)) for i in range(0, 300, 37)]
        num_tokens = sum(len(t) for t in texts)

    a = LanguageModelDataLoader(DS, bs=4, bptt=10, random=False)
    b = LanguageModelDataLoader(DS, bs=4, bptt=10, random=False, device_resident=True)
    assert len(a) == len(b)
    for (xa, ya), (xb, yb) in zip(a, b):
        assert torch.equal(xa, xb) and torch.equal(ya, yb) and xb.is_contiguous()
        assert torch.equal(xb[:, 1:], yb[:, :-1])


import pytest  # noqa: E402


@pytest.mark.gpu
def test_device_resident_loaders_equal_the_dataloader_path_on_the_gpu():
    """VERDICT r1 (row f3): on `cuda`, the device-resident pipelines hand the model exactly the minibatches the reference's
    DataLoader + collater + to_cuda path does (collab, structured data, language model), and the Learner computes the same
    validation loss through either."""
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterDataObj, CollabFilterNet
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataObj, StructuredDataset
    from neuralnetworklibrary_amd.Applications.Text import LanguageModelDataLoader
    from neuralnetworklibrary_amd.General.Core import set_default_device, to_cuda
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device('cuda')
    Learner.verbose = False
    rs = np.random.RandomState(0)
    n = 400
    df = pd.DataFrame({'u': rs.randint(0, 30, n), 'm': rs.randint(0, 20, n), 'r': rs.randint(1, 6, n).astype('float32')})
    labels = [{u: i for i, u in enumerate(sorted(df.u.unique()))}, {m: i for i, m in enumerate(sorted(df.m.unique()))}]
    host = CollabFilterDataObj(df[:300], df[300:], 'u', 'm', 'r', labels, bs=64, num_workers=0)
    dev = CollabFilterDataObj(df[:300], df[300:], 'u', 'm', 'r', labels, bs=64, num_workers=0, device_resident=True)
    nb = 0
    for (xh, yh), (xd, yd) in zip(host.val_dl, dev.val_dl):
        assert xd.is_cuda and yd.is_cuda and torch.equal(to_cuda(xh), xd) and torch.equal(to_cuda(yh), yd)
        nb += 1
    assert nb == 2 and len(dev.train_dl) == len(host.train_dl)
    torch.manual_seed(0)
    net = CollabFilterNet(len(labels[0]), len(labels[1]), 8, [0.8, 5.2])
    lh = Learner('/tmp/nnl_test_dd', host, net, optimizer='Adam').evaluate('val')[0]
    ld = Learner('/tmp/nnl_test_dd', dev, net, optimizer='Adam').evaluate('val')[0]
    assert lh == ld
    learner = Learner('/tmp/nnl_test_dd', dev, net, optimizer='Adam')
    learner.fit(1e-2, 1, wd=1e-4)                                     # an epoch through the device-resident loader trains
    assert learner.evaluate('val')[0] < ld

    xcat = pd.DataFrame(rs.randint(0, 5, (200, 3))); xcont = pd.DataFrame(rs.standard_normal((200, 2)).astype('float32'))
    yv = rs.rand(200).astype('float32')
    tr, va = StructuredDataset(xcat[:150], xcont[:150], yv[:150], 'cont'), StructuredDataset(xcat[150:], xcont[150:], yv[150:], 'cont')
    hs = StructuredDataObj(tr, va, None, None, bs=32, num_workers=0)
    ds = StructuredDataObj(tr, va, None, None, bs=32, num_workers=0, device_resident=True)
    for ((ch, fh), yh), ((cd, fd), yd) in zip(hs.val_dl, ds.val_dl):
        assert cd.is_cuda and torch.equal(to_cuda(ch), cd) and torch.equal(to_cuda(fh), fd) and torch.equal(to_cuda(yh), yd)

    class DS:
        texts = [list(range(i, i + 37)) for i in range(0, 300, 37)]
        num_tokens = sum(len(t) for t in texts)

    a = LanguageModelDataLoader(DS, bs=4, bptt=10, random=False)
    b = LanguageModelDataLoader(DS, bs=4, bptt=10, random=False, device_resident=True)
    for (xa, ya), (xb, yb) in zip(a, b):
        assert xb.is_cuda and torch.equal(to_cuda(xa), xb) and torch.equal(to_cuda(ya), yb)
    set_default_device('cpu')
