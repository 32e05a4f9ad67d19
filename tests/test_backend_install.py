"""backend.install(): rebinding mechanics inside a real torch_geometric (only available in the
build container, where the reference is mounted; skipped elsewhere).  CPU tensors must keep taking
the reference's own path — the backend steps aside, it never computes on the CPU."""
import os
import sys

import pytest
import torch

REF = os.environ.get('PYG_REFERENCE', '/root/reference')
if os.path.isdir(os.path.join(REF, 'torch_geometric')) and REF not in sys.path:
    sys.path.insert(0, REF)

pyg = pytest.importorskip('torch_geometric')


@pytest.fixture()
def installed():
    from pytorch_geometric_amd import backend
    backend.install()
    yield backend
    backend.uninstall()


def test_install_rebinds_every_importer_and_uninstall_restores(installed):
    import torch_geometric.nn.aggr.base as aggr_base
    import torch_geometric.nn.conv.gcn_conv as gcn_mod
    import torch_geometric.utils as U
    import torch_geometric.utils._softmax as sm_mod
    assert installed.is_installed()
    assert pyg.backend.mi355x is installed and pyg.backend.use_mi355x is None
    wrapped = U.scatter
    assert hasattr(wrapped, '__wrapped__')
    # modules that did `from torch_geometric.utils import scatter` see the same wrapper
    assert aggr_base.scatter is wrapped and gcn_mod.scatter is wrapped
    assert sm_mod.scatter is wrapped and sm_mod.segment is U.segment
    assert U.softmax.__wrapped__ is not U.softmax
    n_scatter = sum(1 for _, attr, old in installed._state['rebinds']
                    if old is wrapped.__wrapped__)
    assert n_scatter >= 30, n_scatter  # imported by name all over the package
    orig = wrapped.__wrapped__
    installed.uninstall()
    assert U.scatter is orig and aggr_base.scatter is orig and gcn_mod.scatter is orig
    assert not hasattr(pyg.backend, 'mi355x')
    installed.install()  # the fixture's teardown uninstalls again


def test_cpu_tensors_step_aside_to_the_reference(installed):
    from torch_geometric.nn import GATConv, GCNConv, SAGEConv
    from torch_geometric.utils import scatter, softmax
    g = torch.Generator().manual_seed(0)
    x = torch.randn(10, 8, generator=g)
    ei = torch.randint(0, 10, (2, 40), generator=g)
    idx = torch.randint(0, 4, (10, ), generator=g)
    ref = scatter.__wrapped__(x, idx, 0, 4, 'mean')
    assert torch.equal(scatter(x, idx, 0, 4, 'mean'), ref)
    assert torch.allclose(softmax(x, idx, num_nodes=4),
                          softmax.__wrapped__(x, idx, None, 4, 0))
    for conv in (SAGEConv(8, 4), GCNConv(8, 4), GATConv(8, 4, heads=2)):
        out = conv(x, ei)  # wrapped propagate -> NotImplemented -> reference path
        installed.uninstall()
        ref = conv(x, ei)
        installed.install()
        assert torch.allclose(out, ref, atol=1e-6)


def test_flag_disables_the_backend(installed):
    pyg.backend.use_mi355x = False
    from pytorch_geometric_amd.backend import _enabled
    assert not _enabled()
    pyg.backend.use_mi355x = None
    assert _enabled()
