"""CPU checks of the measurement definitions used by bench.py."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_survey():
    b = load_bench()
    # B_alg(M) = 1.847 GB + 0.556 GB / M: 2.404 (M=1), 1.917 (M=8), 1.856 (M=64)
    assert abs(b.algorithmic_bytes_per_block(1) - 2.403e9) < 2e6
    assert abs(b.algorithmic_bytes_per_block(8) - 1.9165e9) < 2e6
    assert abs(b.algorithmic_bytes_per_block(64) - 1.8557e9) < 2e6
    # blind-rotation launch: 128 bits per block x (6152 B in + 24576 B out) = 3.93 MB per block, + BSK 56.6 MB
    per_block = (b.br_launch_bytes(2 * 128, 768) - b.br_launch_bytes(128, 768))          # units = bits: 128 per block
    assert abs(per_block - 3.93e6) < 0.01e6
    assert abs(b.br_launch_bytes(128, 768) - per_block - 56.62e6) < 0.01e6
    # several launches: the key is streamed once per launch (CTR: eight of 8192 units and one of 2144)
    assert b.br_launch_bytes(8 * 8192 + 2144, 768, 9) == 8 * b.br_launch_bytes(8192, 768) + b.br_launch_bytes(2144, 768)
    assert b.HBM_PEAK_GBS == 8000.0


def test_bench_emits_required_contract_keys_in_source():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"',
                '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"',
                '"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"', '"cores"', '"kind"', '"sample"'):
        assert key in src, key
