"""Recipe for ``oracle/_ref/`` — the REAL reference as test infrastructure on the GPU box.

TEST INFRASTRUCTURE ONLY.  The reference (PyG 2.9.0) is pure Python on this path, so "building"
it is packing its importable package: this script packs ``<reference>/torch_geometric``
(``*.py`` and the ``*.jinja`` templates ``MessagePassing`` renders at class creation) into ONE
archive, ``oracle/_ref/torch_geometric.tar.gz``.  ``oracle/_ref/`` is listed in ``.gitignore`` (no
reference source ever enters the history or sits loose in the tree) but not in ``.gpurunignore``:
like ``lib/libpyg_amd.so`` the archive travels with the snapshot to the GPU box, where
``/root/reference`` does not exist and ``import_reference()`` unpacks it under the temp directory.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may call
``import_reference()``:

* ``tests/test_gpu_reference_install.py`` runs the reference's OWN ``nn.conv.*`` / ``EdgeIndex`` /
  ``utils.*`` on HIP tensors through ``pytorch_geometric_amd.backend.install()`` and compares with
  the reference's CPU results;
* ``tests/test_gpu_reference_suite.py`` runs the reference's own TEST modules for the path (packed
  under ``reference_tests/``) with and without ``install()`` and compares the outcomes;
* ``bench.py`` times the unmodified reference on the host cores (``cpu_baseline.kind =
  "reference"``).

Run by ``__graft_entry__.build()`` whenever ``/root/reference`` is present (the build container);
on the GPU box the staged copy is used as it arrived.
"""
import hashlib
import json
import os
import sys
import tarfile
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get('PYG_REFERENCE', '/root/reference')
DST = os.path.join(HERE, '_ref')
ARCHIVE = os.path.join(DST, 'torch_geometric.tar.gz')
MANIFEST = os.path.join(DST, 'MANIFEST.json')
KEEP = ('.py', '.jinja', '.typed')
# The reference's OWN test modules for the rows of SURVEY.md §8 (+ the conftest that holds their
# fixtures): packed next to the package under `reference_tests/`, so that on the GPU box they run
# against `backend.install()` with their `@withDevice` / `@withCUDA` cases on HIP tensors
# (tests/test_gpu_reference_suite.py).  Data, never edited: any assertion that fails there fails
# as the reference wrote it.
TEST_FILES = ('conftest.py', 'test_edge_index.py', 'test_index.py',
              'nn/models/test_basic_gnn.py')
# ... and every module of these directories: all 60-odd conv layers (the ones without a dedicated
# route ride on `MessagePassing._index_select` + `scatter`), every aggregation, the dense layers,
# the utils
TEST_DIRS = ('nn/conv', 'nn/aggr', 'nn/dense', 'utils', 'nn/models', 'nn/pool', 'nn/norm',
             'nn/functional', 'nn/kge', 'nn/attention', 'nn/unpool', 'explain', 'transforms',
             'nn', 'data', 'sampler', 'metrics')
# ('loader' too was run once — 331 modules, 4,171 cases passed with the backend, 276 s — and left out
# of the standing set: its DataLoader-worker modules double the run time; PYGAMD_REFERENCE_TESTS_MORE
# = a comma-separated list of further directories of <reference>/test for a one-off run)
TEST_DIRS += tuple(d for d in os.environ.get('PYGAMD_REFERENCE_TESTS_MORE', '').split(',') if d)


def _test_files(tsrc):
    found = [t for t in TEST_FILES if os.path.isfile(os.path.join(tsrc, t))]
    for d in TEST_DIRS:
        full = os.path.join(tsrc, d)
        if os.path.isdir(full):
            found += [os.path.join(d, n) for n in sorted(os.listdir(full))
                      if n.endswith('.py') and (n.startswith('test_') or n == 'conftest.py')]
    return list(dict.fromkeys(found))   # (a module named above and found in its directory: once)


def staged_path():
    """The staged archive if it is there, else None."""
    return ARCHIVE if os.path.isfile(ARCHIVE) else None


def stage(force: bool = False):
    """Build container only: pack ``<reference>/torch_geometric`` (``*.py`` + templates) into ONE
    archive under the git-ignored ``oracle/_ref/`` — a built artefact like ``libpyg_amd.so``, not
    loose sources in the tree."""
    src = os.path.join(REF_ROOT, 'torch_geometric')
    if not os.path.isdir(src):
        return staged_path()
    files = []
    for base, dirs, names in os.walk(src):
        dirs[:] = sorted(d for d in dirs if d != '__pycache__')
        for n in sorted(names):
            if n.endswith(KEEP):
                files.append(os.path.relpath(os.path.join(base, n), src))
    tsrc = os.path.join(REF_ROOT, 'test')
    tests = _test_files(tsrc)
    h = hashlib.sha1()
    for rel in files:
        h.update(rel.encode())
        with open(os.path.join(src, rel), 'rb') as f:
            h.update(f.read())
    for rel in tests:
        h.update(('test/' + rel).encode())
        with open(os.path.join(tsrc, rel), 'rb') as f:
            h.update(f.read())
    digest = h.hexdigest()
    if not force and os.path.exists(MANIFEST) and staged_path():
        with open(MANIFEST) as f:
            if json.load(f).get('sha1') == digest:
                return ARCHIVE
    os.makedirs(DST, exist_ok=True)
    for stale in os.listdir(DST):  # loose files of an earlier layout
        path = os.path.join(DST, stale)
        if os.path.isdir(path):
            import shutil
            shutil.rmtree(path, ignore_errors=True)
    tmp = ARCHIVE + '.tmp'
    with tarfile.open(tmp, 'w:gz') as tar:
        for rel in files:
            tar.add(os.path.join(src, rel), arcname=os.path.join('torch_geometric', rel))
        for rel in tests:
            tar.add(os.path.join(tsrc, rel), arcname=os.path.join('reference_tests', rel))
    os.replace(tmp, ARCHIVE)
    with open(MANIFEST, 'w') as f:
        json.dump({'source': src, 'files': len(files), 'sha1': digest}, f)
    return ARCHIVE


def _unpacked_dir():
    """Extract the staged archive once per content hash into the temp directory."""
    with open(MANIFEST) as f:
        digest = json.load(f)['sha1']
    root = os.path.join(tempfile.gettempdir(), f'pyg_reference_{digest[:16]}')
    marker = os.path.join(root, '.complete')
    if not os.path.exists(marker):
        os.makedirs(root, exist_ok=True)
        with tarfile.open(ARCHIVE, 'r:gz') as tar:
            tar.extractall(root)
        with open(marker, 'w') as f:
            f.write(digest)
    return root


def reference_tests():
    """(directory, [test files]) of the reference's own test modules for the path: the mounted
    ``<reference>/test`` in the build container, else the staged copies."""
    if os.path.isdir(os.path.join(REF_ROOT, 'test')):
        root = os.path.join(REF_ROOT, 'test')
    elif staged_path() and os.path.exists(MANIFEST):
        root = os.path.join(_unpacked_dir(), 'reference_tests')
    else:
        raise ImportError('no reference tests: neither /root/reference/test nor a staged archive')
    files = [os.path.join(root, t) for t in _test_files(root)
             if os.path.basename(t) != 'conftest.py']
    return root, files


def import_reference():
    """Put the reference first on ``sys.path`` and import it: the mounted one in the build
    container, else the staged archive (unpacked under the temp directory)."""
    if os.path.isdir(os.path.join(REF_ROOT, 'torch_geometric')):
        path = REF_ROOT
    elif staged_path() and os.path.exists(MANIFEST):
        path = _unpacked_dir()
    else:
        raise ImportError('no /root/reference and no staged archive under oracle/_ref: run '
                          '`python oracle/make_ref.py` in the build container')
    if path not in sys.path:
        sys.path.insert(0, path)
    import torch_geometric
    return torch_geometric


if __name__ == '__main__':
    p = stage(force='--force' in sys.argv)
    print(p or 'reference not found: nothing staged')
