"""bench.py as its own launcher (no GPU needed): `python bench.py --gpus N` with N > 1 and no WORLD_SIZE starts N fresh rank
processes with the torch.distributed environment set, before anything imports torch or touches HIP, and returns their exit code."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = dict(os.environ, **kw)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("n", [3, 8])
def test_plain_multi_gpu_invocation_starts_its_own_ranks(n):
    import torch
    if torch.cuda.is_available():
        pytest.skip("on a GPU box the plain launch is covered by tests/test_bench_gpu.py")
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2"], capture_output=True, text=True, timeout=300,
                       env=_env())
    # no GPU here: every one of the n ranks gets as far as the device check and says so; no CPU fallback, rc != 0
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs MI355X GPUs") == n, r.stderr[-2000:]
    assert "must be launched with" not in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_the_launcher_does_not_initialise_the_gpu_itself():
    """self_launch must run before torch is imported (a process that has initialised HIP must not fork/exec ranks)."""
    import ast
    tree = ast.parse(open(BENCH).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    first_import = min(n.lineno for n in ast.walk(main) if isinstance(n, (ast.Import, ast.ImportFrom)))
    launch_call = min(n.lineno for n in ast.walk(main) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "self_launch")
    assert launch_call < first_import
    top = [a.name for n in tree.body if isinstance(n, ast.Import) for a in n.names]
    assert "torch" not in top and not any(isinstance(n, ast.ImportFrom) for n in tree.body)
    sl = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "self_launch")
    names = {a.name for n in ast.walk(sl) if isinstance(n, ast.Import) for a in n.names}
    assert names <= {"signal", "socket", "subprocess"}, names


def test_rank_environment_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=dict(_env(), WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "does not match --gpus" in r.stderr
