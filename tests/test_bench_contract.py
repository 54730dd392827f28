"""bench.py's accounting (no GPU): the algorithmic bytes of SURVEY.md 8d and the PMC traffic plumbing of the roofline object."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_are_the_survey_figure():
    b = _bench()
    assert b.ALGO_BYTES_PER_HE_MUL == 228_065_280                      # (4R+3W)*30 + (3R+2W)*45 limbs of 512 KiB
    assert (b.LOGN, b.DIM_A, b.DIM_B) == (16, 30, 45)
    assert b.KERNEL_LIMB_PASSES["tensor_mid"] == 7 and b.KERNEL_LIMB_PASSES["keyswitch_mid"] == 5


def test_pmc_traffic_reads_the_committed_summary():
    """roofline.traffic comes from the latest profiles/r*/…pmc_summary.json: 2*FETCH_SIZE + WRITE_SIZE (KiB), scaled to the
    launch-group size.  For the dominant kernel it must sit just above the algorithmic bytes (twiddles), not far above."""
    b = _bench()
    for chunk in (16, 32):
        algo = 7 * 30 * chunk * (8 << 16)
        t, src = b.pmc_traffic("tensor_mid", chunk)
        assert t is not None and 1.0 <= t / algo < 1.15
        assert src["file"].startswith("profiles/r") and "tensor_mid" in src["kernel"]      # the line says where the figure comes from
    for k in ("strided_fwd", "strided_inv", "keyswitch_mid"):
        assert b.pmc_traffic(k, 32)[0] is not None
    t, src = b.pmc_traffic("no_such_kernel", 32)
    assert t is None and "error" in src


def test_pmc_traffic_takes_the_latest_summary_by_number():
    """v10 comes after v9: the summary used must be the one with the highest (round, version), i.e. the current kernels."""
    import glob
    import json
    import re
    b = _bench()
    files = glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_summary.json"))
    latest = max(files, key=lambda p: tuple(int(v) for v in re.search(r"r(\d+)[/\\]v(\d+)_", p).groups()))
    data = json.load(open(latest))
    name = next(k for k in data if "tensor_mid" in k)
    want = int((2 * data[name]["FETCH_SIZE"] + data[name]["WRITE_SIZE"]) * 1024 * 32 / data.get("_chunk", 4))
    assert b.pmc_traffic("tensor_mid", 32)[0] == want


def test_gpus_flag_never_silently_runs_another_rank_count():
    """--gpus N with a launcher that started another WORLD_SIZE is an error (exit status != 0) before any GPU work; with
    WORLD_SIZE unset the parent starts N children itself and hands back their status."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    b = _bench()
    assert b.launch_ranks(2, ["--gpus", "0"]) == 1          # both children refuse --gpus 0; the first failure is the status
    assert b.launch_ranks(2, ["--help"]) == 0


def test_total_batch_partition_is_configs3():
    """BASELINE configs[3]: 512 ciphertexts over 2/4/8 GPUs = 256/128/64 per GPU (block partition of gpqhe_amd/dist.py)."""
    from gpqhe_amd.dist import shard_range
    for world, per in ((2, 256), (4, 128), (8, 64)):
        assert [shard_range(512, world, r)[1] - shard_range(512, world, r)[0] for r in range(world)] == [per] * world
    a = _bench().parse_args(["--gpus", "8", "--total-batch", "512"])
    assert a.total_batch == 512 and a.gpus == 8 and not a.no_scatter_gather


def test_valu_issue_is_computed_from_committed_evidence_not_from_a_literal():
    """bench.py's `valu_issue`: the instruction count per he_mul comes from the newest committed PMC pass, the bound from the newest
    committed probe evaluation (tools/issue_probe + tools/valu_bound.py); both are named in the object, with the commits they were
    taken at, and the object says whether the kernel sources have changed since the probe was evaluated."""
    import json
    b = _bench()
    v = b.valu_issue(7500.0, 1919)
    assert "error" not in v, v
    pmc = json.load(open(os.path.join(ROOT, v["insts_source"]["file"])))
    bound = json.load(open(os.path.join(ROOT, v["bound_source"]["file"])))
    assert v["insts_source"]["file"].startswith("profiles/r") and v["bound_source"]["file"].endswith("valu_bound.json")
    per_group = sum(val["SQ_INSTS_VALU"] * (2 if "strided_pass" in k else 1) for k, val in pmc.items() if not k.startswith("_"))
    assert v["valu_wave_insts_per_he_mul"] == int(per_group / pmc["_chunk"])
    assert v["bound_valu_wave_insts_per_s"] == bound["issue_bound"]["valu_wave_insts_per_s"]
    assert abs(v["frac_of_valu_issue_rate"] - v["valu_wave_insts_per_he_mul"] * 7500.0 / bound["issue_bound"]["valu_wave_insts_per_s"]) < 2e-3
    assert 0.3 < v["frac_of_valu_issue_rate"] < 1.0 and isinstance(v["stale"], bool)
    assert v["insts_source"]["profiled_head"] and v["bound_source"]["probe_head"]
    import inspect
    assert "3.95" not in inspect.getsource(b.valu_issue)


def test_valu_floor_is_the_four_cycle_issue_floor_at_the_given_clock():
    """bench.py's `valu_floor`: SQ_INSTS_VALU per he_mul from the newest committed PMC pass x 4 cycles per
    wave-instruction / (1024 SIMDs x the clock of the same run); frac = floor time x he_mul/s, at most 1."""
    import json
    b = _bench()
    v = b.valu_floor(7472.0, 1913)
    assert "error" not in v, v
    pmc = json.load(open(os.path.join(ROOT, v["insts_source"]["file"])))
    per_group = sum(val["SQ_INSTS_VALU"] * (2 if "strided_pass" in k else 1) for k, val in pmc.items() if not k.startswith("_"))
    insts = per_group / pmc["_chunk"]
    assert v["valu_wave_insts_per_he_mul"] == int(insts) and v["cycles_per_wave_inst"] == 4 and v["simds"] == 1024
    floor_s = insts * 4 / (1024 * 1913e6)
    assert abs(v["frac"] - floor_s * 7472.0) < 1e-3 and 0.8 < v["frac"] <= 1.0          # round 3's operating point: 0.96
    assert b.valu_floor(7472.0, None) is None


import pytest


@pytest.mark.gpu
def test_the_quick_bench_line_prices_kernels_against_ceilings():
    """`python bench.py --quick`: the yardstick of `of_copy_rate` is the best of the library's own plain copies measured in the same
    process, so no kernel may exceed it; the VALU issue floor at the run's clock cannot be exceeded either; the contract keys are there."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-sample", "1", "--quick"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["cpu_baseline"]["bit_exact_vs_gpu"] is True
    rates = {k: v["of_copy_rate"] for k, v in line["kernels"].items()}
    # (1.03: two measurements of the yardstick itself differ by up to 3 %; the forward strided pass runs at 0.93-0.97 of it, the others below 0.9)
    assert rates and all(0 < x <= 1.03 for x in rates.values()), (rates, line["copy_rate"])
    assert sum(rates.values()) / len(rates) <= 0.95, (rates, line["copy_rate"])
    assert line["copy_rate"]["read_only_GBps"] >= line["copy_rate"]["GBps"] * 0.9
    if "valu_floor" in line:                           # needs rocm-smi for the clock
        assert 0.5 < line["valu_floor"]["frac"] <= 1.05, line["valu_floor"]    # (1.05: the clock is rocm-smi's, sampled beside the run, not inside it)
        assert line["roofline"]["bound"] == "valu_issue" and line["roofline"]["declared_bound"] == "hbm"
        # one block of denominators; nothing stale is ever shipped: a probe evaluated on other kernel headers is omitted, not flagged
        den = line["valu_floor"]["denominators"]
        assert den["four_cycles_per_wave_instruction_at_this_clock"] == line["valu_floor"]["frac"]
        assert "stale" not in line.get("valu_issue", {})
        assert ("valu_issue" in line) == ("own_mix_probe_at_its_uncapped_clock" in den)
    assert "stale" not in json.dumps(line)
    assert line["roofline"]["peak"] == 8000.0 and 0 < line["roofline"]["frac"] < 1
    # round 6: both halves of BASELINE's metric ("he_mul/sec + NTT GB/s") and the whole function as top-level scalars the driver's parser keeps
    for key in ("ntt_GBps", "ntt_hbm_frac", "he_mul_whole_per_s", "he_mul_plus_he_rescale_whole_per_s", "per_rank", "affinity"):
        assert key in line, key
    assert 500 < line["ntt_GBps"] < 8000 and abs(line["ntt_hbm_frac"] - line["ntt_GBps"] / 8000) < 1e-3
    assert 1000 < line["he_mul_plus_he_rescale_whole_per_s"] <= line["he_mul_whole_per_s"] * 1.05 < line["value"] * 1.05
    pr = line["per_rank"]
    assert len(pr["ranks"]) == 1 and pr["he_mul_per_s_min"] == pr["he_mul_per_s_max"] == pr["ranks"][0]["he_mul_per_s"]
    assert abs(pr["ranks"][0]["he_mul_per_s"] / line["value"] - 1) < 0.02       # one rank: its own rate IS the aggregate (the closing barrier is a no-op)
    assert isinstance(line["affinity"]["bound"], bool)


def test_top_level_metric_keys_are_in_every_line_shape():
    """the scalars exist (as None) even when their legs do not run: a parser never has to guess (source-level check, no GPU)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"ntt_GBps": None, "ntt_hbm_frac": None, "he_mul_whole_per_s": None, "he_mul_plus_he_rescale_whole_per_s": None' in src
    assert src.index("affinity.bind_to_gpu(") < src.index("    import torch\n    import gpqhe_amd\n")   # placement before torch / HIP in a rank
