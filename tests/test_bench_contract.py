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
        t = b.pmc_traffic("tensor_mid", chunk)
        assert t is not None and 1.0 <= t / algo < 1.15
    for k in ("strided_fwd", "strided_inv", "keyswitch_mid"):
        assert b.pmc_traffic(k, 32) is not None


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
    assert b.pmc_traffic("tensor_mid", 32) == want
