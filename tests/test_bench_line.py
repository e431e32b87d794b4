"""bench.py's stdout contract (CPU): ONE compact JSON line that the driver's 8 KB tail holds completely, with every BASELINE row in it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _verbose_record():
    # a full record of an earlier run (what `bench.py --verbose` prints and what gpurun_out/bench_verbose_*.json holds)
    for line in open(os.path.join(ROOT, "profiles", "r04_bench_cfg2.json")):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no record")


def test_compact_line_fits_the_drivers_tail_and_names_every_baseline_row():
    import bench

    rec = _verbose_record()
    line = bench.compact_line(rec, os.path.join(ROOT, "gpurun_out", "bench_verbose_cfg2_n1.json"))
    text = json.dumps(line)
    assert len(text) <= 4096, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["config"]["workload"].startswith("cfg2") and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (line["ms_per_step"] * 1e-3) / 1e9) < 2.0  # B / t, GB/s
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    # the BASELINE configurations (configs[2..4], cfg5 both as a shard and as stated) and the out-of-cache workload
    for row in ("cfg3", "cfg4", "cfg5", "cfg5_strong", "hbm", "cfg3_rough", "cfg5_rough", "cfg2_theta150", "cfg2_batch8"):
        e = line["configs"][row]
        assert e["us"] > 0 and 0 < e["frac"] < 1 and "dom_us" in e and "prep_ms" in e, row
        assert abs(e["us"] - rec["also"][row]["ms_per_step"] * 1e3) <= 1e-3 * e["us"]
    assert line["verbose"] == "gpurun_out/bench_verbose_cfg2_n1.json"


def test_comm_model_matches_the_design_table():
    """cfg5's C1 (3.7 MB) at N = 2 / 4 / 8 = 12 + 37, 18 + 18.5, 25 + 9.3 us; C2 twice the bytes."""
    import bench

    c1 = 720 * 1280 * 4
    assert abs(bench.comm_model_us(c1, 2) - (12 + 36.9)) < 0.2
    assert abs(bench.comm_model_us(c1, 4) - (18 + 18.4)) < 0.2
    assert abs(bench.comm_model_us(c1, 8) - (25 + 9.2)) < 0.2
    assert abs(bench.comm_model_us(2 * c1, 8) - (25 + 18.4)) < 0.2
    assert abs(bench.comm_model_us(16, 8) - 25.0) < 0.01


def test_band_model_matches_the_design_table():
    """DESIGN.md section 5, row-band partition of cfg5 (1280 wide, 20 halo rows = 102 KB per exchange and direction = 1.0 us on the wire):
    two exchanges at 8 + 1 us, the statistics' all-reduce at alpha(N), the patch gradient's 4 KB at alpha(N) + its wire time."""
    import bench

    assert abs(bench.band_model_us(27.5, 1280, 8) - (27.5 + 2 * (8 + 1.024) + 25.0 + 25.0 + 0.01)) < 0.1
    assert abs(bench.band_model_us(48.0, 1280, 2) - (48.0 + 2 * (8 + 1.024) + 12.0 + 12.04)) < 0.1
    assert abs(bench.band_model_us(27.5, 1280, 8, patch_gradient=False) - (27.5 + 18.05 + 25.0)) < 0.1
    assert bench.band_model_us(85.0, 1280, 1) == 85.0
