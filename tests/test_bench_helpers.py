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


def test_path_bytes_follow_the_units_actually_bootstrapped():
    """the full schedule is 9 x 128 units per block; the CTR schedule's shared second round bootstraps 2144 of the
    medium instance's 8192 units, and only the per-unit stages (KS, blind rotation, trace, scheme switch, LUT tree) shrink"""
    b = load_bench()
    full = b.path_algorithmic_bytes(64, 9 * 8192)
    assert abs(full - 64 * b.algorithmic_bytes_per_block(64)) < 1e3
    shared = b.path_algorithmic_bytes(64, 8 * 8192 + 2144)
    skipped_block_rounds = (8192 - 2144) / 128.0
    assert abs((full - shared) - skipped_block_rounds * 194.52e6) < 1e6
    assert 0.91 < shared / full < 0.93                       # 216 instead of 235 GB/s at the round-3 rate
    assert abs(b.path_algorithmic_bytes(8, 9 * 1024) - 8 * b.algorithmic_bytes_per_block(8)) < 1e3


def test_counter_figures_are_tied_to_the_instruction_stream(tmp_path):
    """profiles/roofline_inputs.json: an entry is used only while the sha256 it records is that of csrc/tx_br_asm_fair.inc;
    every entry of the committed file names source CSVs that exist under profiles/"""
    import hashlib
    import json
    b = load_bench()
    doc = json.load(open(b.ROOFLINE_INPUTS))
    for k, e in doc["entries"].items():
        assert len(e["stream_sha256"]) == 64 and e["kernel"] == "k_blind_rotate5", k
        for part in e["source"].split(" + "):
            assert os.path.exists(os.path.join(ROOT, part)), part
        if "bytes_per_launch" in e:
            assert e["bytes_per_launch"] == (e["fetch_size_kib"] * 2 + e["write_size_kib"]) * 1024
    stream = tmp_path / "stream.inc"
    stream.write_text("v_fma_f64 v[0:1], v[2:3], v[4:5], v[6:7]\n")
    sha = hashlib.sha256(stream.read_bytes()).hexdigest()
    inputs = tmp_path / "inputs.json"
    inputs.write_text(json.dumps({"entries": {"a": {"stream_sha256": sha, "value": 1}, "b": {"stream_sha256": "0" * 64, "value": 2}}}))
    assert list(b.roofline_inputs(str(inputs), str(stream))) == ["a"]
    stream.write_text("v_fma_f64 v[0:1], v[2:3], v[4:5], v[8:9]\n")          # the kernel changed: nothing is reported
    assert b.roofline_inputs(str(inputs), str(stream)) == {}
    assert b.roofline_inputs(str(tmp_path / "missing.json"), str(stream)) == {}


def _run_bench(args, env_drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as a plain command (no RANK): two rank processes are started, rendezvous on 127.0.0.1, shard
    the instance raggedly, gather to rank 0 in block order, and exactly one JSON line comes back (dry run: the launch, shard
    and gather logic with row patterns for ciphertexts, gloo on CPU)"""
    import json
    r = _run_bench(["--gpus", "2", "--backend", "gloo", "--dry-run", "--blocks", "5", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["blocks_per_rank"] == [3, 2] and line["ranks_in_result"] == [0, 1]
    assert line["gathered_in_block_order"] is True and line["dry_run"] is True and line["value"] is None
    # eight ranks, three of them without a block (5 blocks in slabs of one)
    r = _run_bench(["--gpus", "8", "--backend", "gloo", "--dry-run", "--blocks", "5", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["blocks_per_rank"] == [1, 1, 1, 1, 1, 0, 0, 0] and line["gathered_in_block_order"] is True


def test_bench_forced_group_of_one_and_launcher_mismatch():
    import json
    r = _run_bench(["--gpus", "1", "--force-dist", "--backend", "gloo", "--dry-run", "--blocks", "3"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["blocks_per_rank"] == [3]
    # started by a launcher whose world size is not --gpus: one line, non-zero, no rendezvous attempted
    import subprocess
    import sys
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_bench_without_a_gpu_fails_fast_with_one_line():
    """on a node without (enough) GPUs `bench.py --gpus 8` returns at once with one clear line instead of waiting for peers"""
    import time
    import torch
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("this node has eight GPUs")
    t0 = time.time()
    r = _run_bench(["--gpus", "8"])
    assert r.returncode != 0 and time.time() - t0 < 120
    assert ("no HIP device" in r.stderr) or ("need 8 GPUs" in r.stderr)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_vector_instruction_count_comes_from_the_stream(tmp_path):
    """bench.py's vector_issue figure counts the v_ instructions of the generated CMux loop in the tree's instruction stream (between
    the loop label and its backward branch), nothing typed in: a synthetic stream gives its own count, the real one a plausible one"""
    b = load_bench()
    f = tmp_path / "s.inc"
    f.write_text('"  v_mov_b32 v0, 0\\n"\n"L_step%=:\\n"\n"  v_add_f64 v[0:1], v[2:3], v[4:5]\\n"\n"  s_nop 0\\n"\n"  ds_read_b64 v[0:1], v2\\n"\n'
                 '"  v_xad_u32 v1, v2, v3, v4\\n"\n"  s_cbranch_scc1 L_step%=\\n"\n"  v_mov_b32 v9, 0\\n"\n')
    assert b.vector_instructions_per_unit_step(str(f)) == 2
    assert b.vector_instructions_per_unit_step(str(tmp_path / "missing.inc")) is None
    n = b.vector_instructions_per_unit_step()
    assert 2000 < n < 3500
