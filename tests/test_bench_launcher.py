"""`python bench.py --gpus N` must start N ranks by itself (round-1 VERDICT weak #3): the launcher
path — self re-exec under torch.distributed.run, rendezvous on 127.0.0.1, parameter broadcast,
flat-bucket all-reduce, barrier + max-over-ranks timing, `n_gpus` read from the process group —
exercised here with world size 2 on gloo through `--dry-run` (which runs no kernel: the compute
path has no CPU fallback)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e['CUDA_VISIBLE_DEVICES'] = ''  # the launcher check is a CPU/gloo test even on a GPU box
    e['HIP_VISIBLE_DEVICES'] = ''
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args,
                          capture_output=True, text=True, env=e, timeout=timeout, cwd=ROOT)


@pytest.mark.timeout(300)
def test_bench_spawns_its_own_ranks():
    res = _run(['--gpus', '2', '--dry-run', '--steps', '4', '--warmup', '1', '--scale', '0.01'])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout  # exactly ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['dry_run'] is True and out['backend'] == 'gloo'
    assert out['collectives_ok'] is True and out['steps'] == 4


@pytest.mark.timeout(120)
def test_world_size_must_match_gpus():
    # a rendezvous is present (WORLD_SIZE=1) but --gpus says 2: refuse instead of printing n_gpus=1
    res = _run(['--gpus', '2', '--dry-run', '--steps', '1'],
               env={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert res.returncode != 0
    assert 'process group has 1 rank' in res.stderr


@pytest.mark.timeout(120)
def test_compute_modes_refuse_to_run_without_a_gpu():
    res = _run(['--gpus', '1', '--steps', '1', '--warmup', '0', '--scale', '0.001'])
    assert res.returncode != 0
    assert 'no CPU fallback' in res.stderr
