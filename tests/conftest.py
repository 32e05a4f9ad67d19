import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    path = os.path.join(ROOT, 'tests', 'golden', 'golden_v1.pt')
    return torch.load(path, map_location='cpu', weights_only=False)


@pytest.fixture(scope='session')
def golden_preproc():
    path = os.path.join(ROOT, 'tests', 'golden', 'golden_preproc_v1.pt')
    return torch.load(path, map_location='cpu', weights_only=False)


@pytest.fixture(scope='session')
def golden_rgcn():
    path = os.path.join(ROOT, 'tests', 'golden', 'golden_rgcn_v1.pt')
    return torch.load(path, map_location='cpu', weights_only=False)


@pytest.fixture(scope='session')
def golden_aggr():
    path = os.path.join(ROOT, 'tests', 'golden', 'golden_aggr_v1.pt')
    return torch.load(path, map_location='cpu', weights_only=False)


@pytest.fixture(scope='session')
def dev():
    return torch.device('cuda:0')
