"""Recipe for ``oracle/_ref/`` — the REAL reference as test infrastructure on the GPU box.

TEST INFRASTRUCTURE ONLY.  The reference (PyG 2.9.0) is pure Python on this path, so "building"
it is staging its importable package: this script mirrors ``<reference>/torch_geometric``
(``*.py`` and the ``*.jinja`` templates ``MessagePassing`` renders at class creation) into
``oracle/_ref/torch_geometric``.  ``oracle/_ref/`` is listed in ``.gitignore`` (no reference source
ever enters the history) but not in ``.gpurunignore``: like ``lib/libpyg_amd.so`` it travels with
the snapshot to the GPU box, where ``/root/reference`` does not exist.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may put
``oracle/_ref`` on ``sys.path``:

* ``tests/test_gpu_reference_install.py`` runs the reference's OWN ``nn.conv.*`` / ``EdgeIndex`` /
  ``utils.*`` on HIP tensors through ``pytorch_geometric_amd.backend.install()`` and compares with
  the reference's CPU results;
* ``bench.py`` times the unmodified reference on the host cores (``cpu_baseline.kind =
  "reference"``).

Run by ``__graft_entry__.build()`` whenever ``/root/reference`` is present (the build container);
on the GPU box the staged copy is used as it arrived.
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get('PYG_REFERENCE', '/root/reference')
DST = os.path.join(HERE, '_ref')
KEEP = ('.py', '.jinja', '.typed')


def staged_path():
    """``oracle/_ref`` if a staged reference is there, else None."""
    return DST if os.path.isfile(os.path.join(DST, 'torch_geometric', '__init__.py')) else None


def stage(force: bool = False):
    src = os.path.join(REF_ROOT, 'torch_geometric')
    if not os.path.isdir(src):
        return staged_path()
    manifest_path = os.path.join(DST, 'MANIFEST.json')
    files = []
    for base, dirs, names in os.walk(src):
        dirs[:] = sorted(d for d in dirs if d != '__pycache__')
        for n in sorted(names):
            if n.endswith(KEEP):
                files.append(os.path.relpath(os.path.join(base, n), src))
    h = hashlib.sha1()
    for rel in files:
        h.update(rel.encode())
        with open(os.path.join(src, rel), 'rb') as f:
            h.update(f.read())
    digest = h.hexdigest()
    if not force and os.path.exists(manifest_path):
        with open(manifest_path) as f:
            if json.load(f).get('sha1') == digest and staged_path():
                return DST
    out = os.path.join(DST, 'torch_geometric')
    shutil.rmtree(out, ignore_errors=True)
    for rel in files:
        dst = os.path.join(out, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(src, rel), dst)
    with open(manifest_path, 'w') as f:
        json.dump({'source': src, 'files': len(files), 'sha1': digest}, f)
    return DST


def import_reference():
    """Put the staged reference (or the mounted one) first on ``sys.path`` and import it."""
    path = staged_path()
    if path is None and os.path.isdir(os.path.join(REF_ROOT, 'torch_geometric')):
        path = REF_ROOT
    if path is None:
        raise ImportError('no staged reference under oracle/_ref and no /root/reference: run '
                          '`python oracle/make_ref.py` in the build container')
    if path not in sys.path:
        sys.path.insert(0, path)
    import torch_geometric
    return torch_geometric


if __name__ == '__main__':
    p = stage(force='--force' in sys.argv)
    print(p or 'reference not found: nothing staged')
