"""The bench line's contract (no GPU): the newest committed end-of-round line (profiles/r*_bench_cfgB.json, printed by
`python bench.py` on an MI355X) carries every field the driver and the judge read, with consistent arithmetic; and the launcher
logic of `--gpus N` (self-spawn, refusal of a mismatched world) works without a GPU."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_a_world_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in (out.stderr + out.stdout)


def test_bench_gpus_n_spawns_n_ranks():
    """No WORLD_SIZE in the environment: `--gpus 2` re-executes itself under torch.distributed.run with two ranks (which then stop
    at the GPU check here: the launcher ran, both ranks started)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=600)
    text = out.stderr + out.stdout
    assert out.returncode != 0
    assert text.count("bench.py needs an MI355X") >= 2 or "local_rank: 1" in text or "rank: 1" in text, text[-2000:]


def test_committed_bench_line_has_the_contract_fields():
    path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_cfgB.json")))[-1]
    with open(path) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    if line["dtype"] == "f32":          # round 3 on: the reference's arithmetic is the headline, bf16 rides beside it
        assert line["value_bf16"] > 0 and line["value_f32_library_gemm"] > 0 and line["value_with_attention_output"] > 0
        assert line["roofline"]["kernel"].startswith(("sparse_attn_x3_kernel", "sparse_attn_x3p_kernel"))
        assert line["roofline_bf16"]["kernel"].startswith("sparse_attn_mfma_kernel")
        assert line["vit_bf16"]["value"] > 0 and line["vit_f32"]["value"] > 0 and line["vit_bf16"]["roofline"]["bound"] == "mfma"
        line = dict(line, dtype="bf16", value_f32=line["value"], roofline=line["roofline_bf16"], roofline_f32=line["roofline"])
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
    assert line["roofline_f32"]["kernel"].startswith(("sparse_attn_x3_kernel", "sparse_attn_x3p_kernel"))     # the kernel the fp32 model dispatches
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["unit"] == "slides/s"
    sweep = cpu["thread_sweep"]
    assert cpu["value"] == max(sweep.values()) and str(cpu["cores"]) == max(sweep, key=sweep.get)
