"""The line `bench.py` prints is a contract with the driver (one JSON object on stdout, rank 0): the keys it reads, their
types, and what this round added (the spot check, the trial scalars).  A short run with the driver's own flag set."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "UWSPR_OPTIONS"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--no-sweep", "--no-lazy", "--no-host-legs", "--trial-steps", "12"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]                      # ONE line on stdout
    d = json.loads(lines[0])
    assert d["metric"].startswith("2-min WSPR frames decoded/sec") and d["unit"] == "frames/s"
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert isinstance(d["value"], float) and d["value"] > 0
    assert abs(d["value"] - 256 * 6 / (d["ms_per_step"] * 1e-3 * 6)) < 1e-6 * d["value"]      # value = frames / time
    cfg = d["config"]
    assert "configs[1]" in cfg["workload"] and "model" not in cfg and cfg["frames_per_gpu"] == 256
    rf = d["roofline"]
    assert rf["bound"] == "valu_fp32_nofma" and rf["unit"] == "Top/s" and rf["peak"] == 78.6
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.2 < rf["frac"] < 1.0
    assert "traffic" in rf and rf["kernel_ms_per_step"] > 0 and rf["launches_per_step"] >= 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    sc = d["parity_spot_check"]
    assert sc["equal"] is True and sc["frames"] == 8 and sc["mismatches"] == []
    # the (form, streams) trial: its table and the choice as top-level scalars, its length independent of --steps
    assert d["trial_steps"] == 12 and d["chosen_sched"] in ("staged", "fused") and d["chosen_streams"] in (1, 2, 3)
    for f in ("staged", "fused"):
        for n in (1, 2, 3):
            assert d["trial_%s_x%d_ms" % (f, n)] > 0
    best = min(d["trial_%s_x%d_ms" % (f, n)] for f in ("staged", "fused") for n in (1, 2, 3))
    assert d["trial_%s_x%d_ms" % (d["chosen_sched"], d["chosen_streams"])] <= 1.02 * best
