"""The committed bench line (profiles/r04_bench.json, written by bench.py on an MI355X through tools/collect_profiles.sh) carries
every key the driver's contract names, and its own numbers are consistent with each other.  No GPU needed."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))


def test_contract_keys():
    d = load()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")


def test_line_is_self_consistent():
    d = load()
    shape_samples = 512 * 64000
    assert abs(d["value"] - d["n_gpus"] * shape_samples / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    # algorithmic bytes: 4 + 4 (H + 2) / hop per sample, over the synth kernel's average launch
    assert abs(r["algorithmic_bytes_per_launch"] - shape_samples * (4 + 4 * 102 / 128)) < 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]
    # the clock: probe reading, the from-idle pass beside the sustained one
    assert 1.5 < d["clock_ghz"] < 2.6
    idle = d["clock_settle"]["from_idle"]
    assert idle["ms_per_step"] > 0 and d["clock_settle"]["continuous_load_ms_before_warmup"] >= 60.0
    # (only the synth kernel is timed inside the timed region; the others come from K more steps with events around every launch,
    #  which slow the stream a little: their sum may exceed the step by that much)
    assert d["kernel_ms"]["osc_frame_synth"] < d["ms_per_step"] and abs(sum(d["kernel_ms"].values()) / d["ms_per_step"] - 1.0) < 0.06
    for k in ("cfg1", "cfg2", "cfg3", "musical", "live_callback", "train_step"):
        assert k in d["configs"], k


def test_rocprof_timed_region_agrees_with_the_bench_events():
    """profiles/r04_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same command): the average over the timed region's
    launches of the dominant kernel is within 3 % of the HIP-event average of the plain run on the same box.  (An event pair is not
    a barrier: its reading includes what was left of the PRECEDING kernel when the first event was reached -- little for the synth
    kernel, which follows the 13 us scan; the persistent noise kernel's tail for the totals kernel: 15 % allowed there.)"""
    d = load()
    rows = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r04_kernel_stats.csv")))}
    synth = rows["osc_chunk_synth_kernel<13, false>"]
    assert abs(float(synth["timed_region_avg_ns"]) * 1e-6 / d["kernel_ms"]["osc_frame_synth"] - 1.0) < 0.03
    assert abs(float(rows["osc_chunk_totals_kernel<13>"]["timed_region_avg_ns"]) * 1e-6 / d["kernel_ms"]["osc_frame_totals"] - 1.0) < 0.15
