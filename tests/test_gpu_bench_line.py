"""bench.py on the GPU box, short runs: the strong-scaling branch (BASELINE config 5's sharding: one global list, rank r
decodes shard_range(T, r, world)) exercised on ONE GPU before an 8-GPU node ever sees it, and the default (weak) line's
contract keys.  The driver's own timing runs use the defaults; these runs use few steps and skip the CPU leg."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*flags):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu",
                        "--no-extras"] + list(flags), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_strong_scaling_branch_on_one_gpu(built):
    d = run_bench("--scaling", "strong", "--total-units", "8192")
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    c = d["config"]
    assert c["bit_exact"] is True and c["units_per_gpu"] == 8192 and c["units_on_frame_parallel_path"] == 1.0
    assert abs(d["value"] - 8192 * c["unit_bytes"] / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01
    assert d["roofline"]["frac"] < 1.0 and d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.02


def test_weak_line_keys(built):
    d = run_bench("--units", "1024")
    assert d["scaling"] == "weak" and d["unit"] == "MB/s" and d["dtype"] == "u8" and d["vs_baseline"] is None
    assert d["config"]["bit_exact"] is True and "workload" in d["config"] and d["roofline"]["bound"] == "hbm"
