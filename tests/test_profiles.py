"""The committed counter pass (profiles/pmc_current.json) that bench.py quotes as `roofline.traffic` must belong to the
sweep kernels that ship: its signature is recomputed here from the current sources and the layout it recorded.  A change
to the sweep (csrc/spmv_device.h, csrc/spmv.hip) without a new PMC pass (tools/profile_round.sh + tools/summarize_profiles.py)
fails this test instead of letting bench.py silently drop - or worse, keep - stale traffic."""
import importlib.util
import json
import os

from conftest import REPO


def test_committed_counter_pass_matches_the_shipped_sweep_sources():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_current.json")))
    assert pmc["config"] == {"N": 10_000_000, "nnz_row": 64, "n_gpus": 1}          # BASELINE's metric configuration
    assert pmc.get("layout") and pmc.get("signature")
    assert bench.kernel_signature(pmc["layout"]) == pmc["signature"]
    hits = [v for k, v in pmc["kernels"].items() if "spmv_tcoow_kernel" in k]
    assert hits and 4.0e9 <= hits[0]["hbm_bytes_per_launch"] <= 5.5e9                # algorithmic: 4.00 GB per launch
