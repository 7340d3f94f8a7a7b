"""The bench line's contract (no GPU): the committed end-of-round line (profiles/r02_bench_cfgB.json, printed by `python bench.py`
on an MI355X) carries every field the driver and the judge read, with consistent arithmetic."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r02_bench_cfgB.json")) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["unit"] == "slides/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and line["n_gpus"] == 1
    assert "slides/sec" in str(base.get("metric", "")) and line["metric"] == "slides/sec"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert "N=32768" in line["config"]["workload"] and "D=768" in line["config"]["workload"]
    # value is the whole-job rate of EXACTLY `steps` timed steps
    assert abs(line["value"] - line["n_gpus"] * 1e3 / line["ms_per_step"]) / line["value"] < 2e-3
    # both arithmetics in one driver-run line
    assert line["dtype"] == "bf16" and line["value_f32"] > 0 and line["value_f32_library_gemm"] > 0
    assert line["value_with_attention_output"] > 0
    for name in ("roofline", "roofline_f32"):
        r = line[name]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "us_per_launch", "algorithmic_bytes"):
            assert key in r, (name, key)
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["us_per_launch"] * 1e-6) / 1e9) / r["achieved"] < 2e-3
        assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes"]
    assert line["roofline"]["kernel"].startswith("sparse_attn_mfma_kernel")
    assert line["roofline_f32"]["kernel"].startswith("sparse_attn_x3_kernel")     # the kernel the fp32 model dispatches
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["unit"] == "slides/s"
    sweep = cpu["thread_sweep"]
    assert cpu["value"] == max(sweep.values()) and str(cpu["cores"]) == max(sweep, key=sweep.get)
