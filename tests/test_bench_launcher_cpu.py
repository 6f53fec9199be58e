"""bench.py --gpus N: plain `python bench.py --gpus N` must start the N ranks itself as a CHILD torchrun job (the
reference's nn.DataParallel, pretrain.py:63-65, replaced by one process per GPU) and relay rank 0's line -- or fail loudly. It must never
print an n_gpus-1 line for an N-GPU request. The launcher logic runs here with a stub in place of subprocess.run; no GPU, no ranks."""
import importlib.util
import io
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_mod_launcher', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_builds_the_torchrun_child_and_relays_rank0s_line():
    bench = _bench()
    seen = {}

    def fake_run(cmd, stdout=None, text=None, env=None):
        seen['cmd'], seen['env'] = cmd, env
        line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": 4, "rccl_ranks": 4})
        return types.SimpleNamespace(returncode=0, stdout='NCCL version banner\n' + line + '\n')

    out = io.StringIO()
    rc = bench.launch_ranks(4, ['--gpus', '4', '--steps', '5', '--warmup', '2'], device_count=8, run=fake_run, out=out)
    assert rc == 0
    cmd = seen['cmd']
    assert cmd[0] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--nnodes') + 1] == '1'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert 1024 < int(cmd[cmd.index('--master-port') + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '5', '--warmup', '2']            # the ranks get the same flags
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 4                     # ONE JSON line, the banner is dropped


def test_launcher_refuses_more_ranks_than_gpus_and_a_line_for_another_n():
    bench = _bench()
    called = []
    out = io.StringIO()
    rc = bench.launch_ranks(2, ['--gpus', '2'], device_count=1, run=lambda *a, **k: called.append(1), out=out)
    assert rc != 0 and not called and out.getvalue() == ''

    def one_rank_line(cmd, **kw):
        return types.SimpleNamespace(returncode=0, stdout=json.dumps({"n_gpus": 1}) + '\n')
    rc = bench.launch_ranks(2, ['--gpus', '2'], device_count=2, run=one_rank_line, out=out)
    assert rc != 0 and out.getvalue() == ''

    def failing(cmd, **kw):
        return types.SimpleNamespace(returncode=7, stdout='')
    assert bench.launch_ranks(2, ['--gpus', '2'], device_count=2, run=failing, out=out) == 7

    def failing_after_the_line(cmd, **kw):                                             # rank 0 printed, then a rank died: the line is withheld
        return types.SimpleNamespace(returncode=9, stdout=json.dumps({"n_gpus": 2, "value": 1.0}) + '\n')
    assert bench.launch_ranks(2, ['--gpus', '2'], device_count=2, run=failing_after_the_line, out=out) == 9 and out.getvalue() == ''
    assert isinstance(bench._kfd_gpu_count(), int) and bench._kfd_gpu_count() >= 0      # counted without a HIP call (0 on this GPU-less box)
    assert bench.launch_ranks(2, ['--gpus', '2'], device_count=2, run=lambda cmd, **kw: types.SimpleNamespace(returncode=0, stdout='no json\n'), out=out) != 0


def test_gpus_2_on_this_gpuless_box_fails_loudly():
    """The real thing, end to end: no stub. This container has no GPU, so `--gpus 2` must exit non-zero without a JSON line --
    and so must a torchrun environment whose WORLD_SIZE disagrees with the flag."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')})
    assert r.returncode != 0 and 'refusing' in r.stderr and '{' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'))
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr and '{' not in r.stdout
