"""GPU: the N > 1 control flow of bench.py — what the driver launches as `torch.distributed.run --nproc-per-node N bench.py --gpus N` —
rehearsed on ONE card: two ranks over gloo sharing device 0 (MP_BENCH_REHEARSE=1: the library's collectives staged through the
host).  The numbers mean nothing; every leg of the line must run, the filter must be ONE filter sharded over the ranks (log-ML
against the Kalman filter), the exchange `value` is measured with must be the library's default — owner-keeps, the single filter's
multiset of parents —, and the split multinomial must be there as a supplementary leg."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_bench_line_on_one_card():
    env = dict(os.environ, MP_BENCH_REHEARSE="1", OMP_NUM_THREADS="2")
    env.pop("MP_SHARD_EXCHANGE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--particles", "65536", "--repeats", "2"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]          # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["particles_total"] == 2 * 65536 and d["config"]["exchange"] == "owned"
    assert d["value"] > 0 and abs(d["value"] - 2 * 65536 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["log_ml_abs_err_vs_kalman"] < 0.2           # one filter of 131072 particles over both ranks
    assert d["c5"]["particles_total"] == 2 * (1 << 21) and d["c5"]["exchange"] == "owned" and d["c5"]["ms_per_step"] > 0
    sp = d["split_multinomial"]
    assert sp.get("exchange") == "split" and sp["ms_per_step"] > 0, sp
    assert abs(sp["log_ml"] - d["log_ml"]) < 0.5         # the same law: another seeded stream, the same filter
