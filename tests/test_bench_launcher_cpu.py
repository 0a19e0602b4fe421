"""bench.py --gpus N without an external launcher: the parent spawns one rank per GPU with the
torch.distributed environment set, fails loudly when there are fewer GPUs than ranks, and hands
back the worst exit code.  No GPU needed: the ranks here are stub processes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _stub(tmp_path, body):
    p = tmp_path / "rank_stub.py"
    p.write_text(body)
    return [sys.executable, str(p)]


def test_spawns_one_rank_per_gpu_with_the_rendezvous_environment(tmp_path):
    argv = _stub(tmp_path, "import os, json\n"
                           "keys = ['RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY']\n"
                           "open(os.path.join(%r, 'rank%%s.json' %% os.environ['RANK']), 'w').write(json.dumps({k: os.environ[k] for k in keys}))\n"
                           % str(tmp_path))
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(4, child_argv=argv, ndev=4)
    assert e.value.code == 0
    envs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(4)]
    assert [int(v["RANK"]) for v in envs] == [0, 1, 2, 3]
    assert [int(v["LOCAL_RANK"]) for v in envs] == [0, 1, 2, 3]
    assert all(v["WORLD_SIZE"] == "4" and v["MASTER_ADDR"] == "127.0.0.1" for v in envs)
    assert len({v["MASTER_PORT"] for v in envs}) == 1 and envs[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_fewer_gpus_than_ranks_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.delenv("SDR_BENCH_BACKEND", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(8, child_argv=_stub(tmp_path, "raise SystemExit(0)\n"), ndev=1)
    assert "only 1 GPU(s) visible" in str(e.value.code)
    # the rehearsal backend lets ranks share a device, but not run without any
    monkeypatch.setenv("SDR_BENCH_BACKEND", "gloo")
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(2, child_argv=_stub(tmp_path, "raise SystemExit(0)\n"), ndev=0)
    assert "needs a GPU" in str(e.value.code)


def test_a_failing_rank_ends_the_others_and_sets_the_exit_code(tmp_path):
    argv = _stub(tmp_path, "import os, sys, time\n"
                           "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                           "time.sleep(60)\n")
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(3, child_argv=argv, ndev=3)
    assert e.value.code == 3


def test_bare_command_without_gpus_says_so():
    """The driver's form of the command, here where no GPU is visible: a clear message, not a hang
    and not a silent n_gpus: 1 line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=120)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("GPUs present: the real launch is exercised by the GPU tests")
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""


def test_strong_scaling_plan_covers_the_64_channels():
    from sdr_amd import sharding
    for world in (1, 2, 4, 8):
        plan = sharding.plan_channels(64, world)
        assert sum(c for _, c in plan) == 64 and all(c == 64 // world for _, c in plan)
    assert bench.WORKLOADS["c4chain"][5] == "strong" and bench.WORKLOADS["c4"][5] == "strong"
    assert bench.WORKLOADS["c2"][5] == "weak"


def test_kernel_flop_counts():
    # direct form: 2 flops per kept tap per real lane
    assert bench.kernel_flops_per_sample("fir_direct<R=4>", 31, True, 31) == 124.0
    assert bench.kernel_flops_per_sample("fir_direct<R=8>", 127, False, 64) == 128.0
    # overlap-save: 390 144 flops per 4096-point tile, 3968 new complex samples at 127 taps
    assert abs(bench.kernel_flops_per_sample("fir_fft4096<COMPLEX>", 127, True, 127) - 390144 / 3968) < 1e-9
    assert abs(bench.kernel_flops_per_sample("fir_fft4096<REAL>", 255, False, 255) - 390144 / (2 * 3840)) < 1e-9
    c3 = bench.kernel_flops_per_sample("fir_fft4096<COMPLEX>", 4095, True, 4095)
    assert 200 < c3 < 210
