"""pytest plugin (``-p tests._install_plugin``): ``backend.install()`` before the reference's OWN
test modules are collected — they then run their ``@withDevice`` / ``@withCUDA`` cases on HIP
tensors through this package's kernels (tests/test_gpu_reference_suite.py)."""


def pytest_configure(config):
    import torch_geometric  # noqa: F401  (the staged / mounted reference, first on sys.path)
    from pytorch_geometric_amd import backend
    backend.install()
    config._pygamd_installed = True


def pytest_unconfigure(config):
    if getattr(config, '_pygamd_installed', False):
        from pytorch_geometric_amd import backend
        backend.uninstall()
