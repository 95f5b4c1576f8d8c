"""bench.py's contract with the driver, checked without a GPU: the ONE printed line stays small enough for the driver's
capture (round 3 lost its record to a 29.8 KB line) and carries the keys the driver and the judge read; and a plain
`python bench.py --gpus N` (no launcher, WORLD_SIZE unset) turns itself into the N-rank torch.distributed.run job."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _full_record():
    row = dict(bound="mfma", achieved=739.0, peak=2500.0, unit="TFLOP/s", frac=0.2956, frac_of_mfma_peak=0.2956,
               frac_of_hbm_peak=0.1, launches=4734, avg_launch_us=99.5, flop_per_launch=7.35e10,
               algorithmic_bytes_per_launch=1.9e8, tflops=739.0, algorithmic_gbps=1900.0, traffic=2.5e8,
               traffic_vs_algorithmic=1.34, share_of_probe_video=0.21)
    fams = {"kernel family %d with a long descriptive name" % i: dict(row) for i in range(60)}
    return {
        "metric": "videos/min (16f x 512x512 SD1.5+AnimateDiff-v3 arch, 30-step DDIM, 18 guided, MotionClone guidance)",
        "value": 34.2, "unit": "videos/min", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1753.7,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "BASELINE config 2 (t2v_object-style): 16 frames, 512x512, UNet only " + "x" * 80,
                   "videos_per_gpu": 20, "videos_in_flight_per_gpu": 3, "parallelism": "replicas x1"},
        "sec_per_guided_step": 0.19, "sec_per_plain_step": 0.13, "sec_per_denoise_step": 0.167,
        "e2e_tflops_per_gpu": 714.6, "e2e_frac_of_mfma_peak": 0.2858,
        "roofline": dict(row, kernel="gemm5<256x320> DENSE"), "roofline_note": "n" * 900,
        "roofline_by_kernel": fams, "roofline_traffic_by_shape": [dict(row) for _ in range(80)],
        "hbm_footprint": dict(peak_allocated_gib=18.3, peak_reserved_gib=63.4, videos_in_flight=3, note="n" * 300),
        "vae": dict(decode_sec_per_video=0.064, note="n" * 200), "graphs": dict(enabled=True, note="n" * 300),
        "eager": dict(videos_per_min=27.9, sec_per_video=2.14, identical_to_graph_path=True, note="n" * 100),
        "reference_gpu_baseline": dict(videos_per_min=1.5, note="n" * 300),
        "cpu_baseline": dict(value=0.417, unit="videos/min (BASELINE config 1: 16f x 256x256, schedule (10,5,0.3), UNet only)",
                             cores=32, cpu_model="AMD EPYC 9575F 64-Core Processor", kind="port", sample="s" * 250,
                             extraction_s=3.4, plain_step_s=13.2, guided_step_s=14.8),
    }


def test_the_printed_line_is_small_and_complete(bench):
    res = _full_record()
    assert len(json.dumps(res)) > 20000                      # the record that overflowed the driver's capture
    line = bench.compact_line(res, "profiles/r05_bench_detail.json")
    assert len(line) <= 4000 and "\n" not in line
    got = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "e2e_frac_of_mfma_peak", "roofline", "cpu_baseline", "detail"):
        assert k in got, k
    assert got["config"]["workload"].startswith("BASELINE config 2") and "model" not in got["config"]
    r = got["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "traffic_vs_algorithmic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = got["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 32 and c["value"] > 0 and "cpu_model" in c and "sample" in c
    assert "roofline_by_kernel" not in got and "roofline_traffic_by_shape" not in got


def test_the_timed_regime_row_is_built_from_the_committed_kernel_trace_and_pinned_in_the_line(bench):
    """roofline_timed (round 5): per-launch duration of the TIMED regime from a rocprofv3 kernel trace of the same command
    (profiles/kernel_durations_timed.json), algorithmic FLOP per launch from the probe video; split-K and GEGLU launches of a
    structure belong to its family, as in the probe."""
    fam = "gemm5<256x320> DENSE"
    row = lambda n, fl: dict(launches=n, flop_per_launch=fl, share_of_probe_video=0.2)   # noqa: E731
    roof_all = {fam: row(3000, 60e9), fam + " split-K + reduce": row(1000, 100e9), "gemm5<256x320> CONV_S1": row(1200, 300e9),
                "attn_fwd_ring self d=40 Nk=4096": row(300, 700e9)}
    prof = dict(regime="test", total_kernel_s=2.0, overlap=1.4, code="abc1234", kernels={
        "gemm5_kernel<0, 0, 0, 256, 320, 8, 4>": dict(calls=30000, avg_us=20.0),
        "gemm5_kernel<0, 1, 0, 256, 320, 8, 4>": dict(calls=10000, avg_us=40.0),     # fused GEGLU: same family
        "gemm5_kernel<1, 0, 0, 256, 320, 8, 4>": dict(calls=1000, avg_us=260.0),
        "gn_apply_kernel": dict(calls=999, avg_us=25.0)})
    assert bench.family_of_traced_kernel("void mc::gemm5_kernel<3, 0, 0, 256, 320, 8, 4>(mc::GemmParams)") == "gemm5<256x320> CONV_UP"
    assert bench.family_of_traced_kernel("gemm5_kernel<0, 0, 0, 256, 160, 4, 3>") == "gemm5<256x160 x2 per CU> DENSE"
    assert bench.family_of_traced_kernel("gemm4_kernel<20, true, 1>") == "gemm4<K=320 streaming> LayerNorm + DENSE"
    assert bench.family_of_traced_kernel("gn_partial_kernel") is None
    assert bench.family_of_traced_kernel("void mc::gemm6_kernel<1, 0, 0, 0>(mc::G6Args)") == "gemm6<256x320 tile loop> DENSE"
    # round 6: a trace is only used if it was taken from the loaded library (source stamp), on this device, with this packing
    stale = bench.timed_roofline(roof_all, prof)
    assert stale["achieved"] is None and stale["frac"] is None and "stale" in stale["reason"]
    prof.update(lib_stamp=bench.loaded_lib_stamp(), device=None, lanes=2, batch=5)      # (no GPU here: device name None on both sides)
    other = bench.timed_roofline(roof_all, prof, 3, 1)
    assert other["achieved"] is None and "lanes" in other["reason"]
    line = json.loads(bench.compact_line(dict(_full_record(), roofline_timed=stale), "d.json"))
    assert line["roofline_timed"]["achieved"] is None and "reason" in line["roofline_timed"]
    rt = bench.timed_roofline(roof_all, prof, 2, 5)
    assert rt["kernel"] == fam and abs(rt["avg_launch_us"] - 25.0) < 1e-9
    assert abs(rt["flop_per_launch"] - 70e9) < 1 and abs(rt["achieved"] - 70e9 / 25.0 / 1e6) < 1e-6
    assert abs(rt["frac"] - rt["achieved"] / 2500.0) < 1e-12 and abs(rt["share_of_kernel_time"] - 0.5) < 1e-9 and rt["overlap"] == 1.4
    assert bench.timed_roofline(roof_all, None) is None and bench.timed_roofline({}, prof, 2, 5) is None
    res = _full_record()
    res["roofline_timed"] = rt
    got = json.loads(bench.compact_line(res, "profiles/r05_bench_detail.json"))
    for k in ("kernel", "achieved", "frac", "avg_launch_us", "overlap", "share_of_kernel_time", "source"):
        assert k in got["roofline_timed"], k


def test_a_line_is_printed_whatever_the_strings_are(bench):
    """compact_line never raises after the benchmark has run (round-4 advice): oversized strings are shed progressively and
    the contract keys survive"""
    res = _full_record()
    res["metric"] = "m" * 3000
    res["config"]["workload"] = "w" * 3000
    res["cpu_baseline"] = {"error": "RuntimeError: " + "x" * 5000}
    res["roofline"]["kernel"] = "k" * 3000
    line = bench.compact_line(res, "profiles/r05_bench_detail.json")
    assert len(line) <= 4000
    got = json.loads(line)
    assert got["value"] == 34.2 and got["unit"] == "videos/min" and got["metric"].startswith("m") and got["n_gpus"] == 1


def test_bench_gpus_8_is_the_eight_rank_job_of_config_3(bench, monkeypatch):
    """BASELINE config 3 = `bench.py --gpus 8`: without a launcher it becomes the 8-rank torch.distributed.run job on
    127.0.0.1 (one rank per GPU, replicas); with the driver's launcher environment the world size must match."""
    seen = {}

    def fake_execv(exe, argv):
        seen["argv"] = list(argv)
        raise SystemExit(0)
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    with pytest.raises(SystemExit):
        bench.main()
    a = seen["argv"]
    assert a[a.index("--nproc-per-node") + 1] == "8" and a[a.index("--master-addr") + 1] == "127.0.0.1" and "--nnodes=1" in a
    assert a[a.index(os.path.join(ROOT, "bench.py")) + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=4" in str(e.value) and "--gpus 8" in str(e.value)


def test_a_failed_baseline_leg_still_gives_a_parsable_line(bench):
    res = _full_record()
    res["cpu_baseline"] = {"error": "RuntimeError: " + "x" * 500}
    res["roofline"] = None
    got = json.loads(bench.compact_line(res, None))
    assert got["value"] == 34.2 and "error" in got["cpu_baseline"] and "roofline" not in got


def test_gpus_n_without_a_launcher_spawns_its_own_ranks(bench, monkeypatch):
    seen = {}

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, list(argv)
        raise SystemExit(0)
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit):
        bench.main()
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in a and a[a.index("--nproc-per-node") + 1] == "2" and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert int(a[a.index("--master-port") + 1]) > 0
    i = a.index(os.path.join(ROOT, "bench.py"))
    assert a[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]


def test_a_mismatched_launcher_is_an_error_not_an_assert(bench, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=4" in str(e.value)


def test_rounds_of_videos_in_flight(bench):
    """--steps videos in rounds of up to --inflight lanes, never leaving a single video for the last round when avoidable"""
    assert bench.plan_rounds(20, 3) == [3, 3, 3, 3, 3, 3, 2]
    assert bench.plan_rounds(4, 3) == [2, 2] and bench.plan_rounds(7, 3) == [3, 2, 2] and bench.plan_rounds(5, 3) == [3, 2]
    assert bench.plan_rounds(1, 3) == [1] and bench.plan_rounds(6, 2) == [2, 2, 2] and bench.plan_rounds(0, 3) == []
    for n in range(1, 40):
        for lanes in (1, 2, 3, 4):
            r = bench.plan_rounds(n, lanes)
            assert sum(r) == n and max(r) <= lanes and (1 not in r or n == 1 or lanes <= 2 and n % lanes == 1 or lanes == 1)


def test_probe_prices_the_fused_norm_gemm_by_its_own_arguments():
    """the roofline probe takes flop / bytes from the C-ABI arguments of the call it brackets: mc_norm_gemm_f16"""
    from motionclone_amd import probe
    M, N, K = 131072, 960, 320
    args = (1, 2, 3, None, M, N, K, K, N, 1, 4, 5, None, 4096, 16, 1e-5, 6, None, 0, 0)
    fam, fl, nb, shape = probe._cost("mc_norm_gemm_f16", args)
    assert fam == "gemm4<K=320 streaming> LayerNorm + DENSE" and fl == 2.0 * M * N * K
    assert nb == 2.0 * (M * K + N * K + M * N) and shape == (1, M, N, K, False)
    fam, fl, nb, shape = probe._cost("mc_norm_gemm_f16", args[:9] + (2,) + args[10:18] + (0x200, 0))
    assert fam.startswith("gemm4<K=320 streaming> GroupNorm") and nb == 2.0 * (2 * M * K + N * K + M * N // 2) and shape[-1] is True
    assert probe._gemm_name(56, 0) == "gemm5<256x160 x2 per CU> DENSE" and probe._gemm_name(151, 1).endswith("CONV_S1 split-K + reduce")


def test_automatic_packing_divides_the_timed_videos(bench):
    """round 6: without --inflight / --batch the bench packs config 2 as two lanes x up to five batched videos, and only where
    lanes x batch divides --steps (every round a full one: EXACTLY --steps videos are timed); other shapes and SparseCtrl keep
    one video per lane"""
    assert bench.auto_packing(20, 16, 512, False) == (2, 5)        # the driver's command
    assert bench.auto_packing(6, 16, 512, False) == (2, 3)         # the no-flag default
    assert bench.auto_packing(8, 16, 512, False) == (2, 4)
    assert bench.auto_packing(7, 16, 512, False) == (3, 1)
    assert bench.auto_packing(4, 16, 512, False) == (2, 2)
    assert bench.auto_packing(6, 16, 512, True) == (3, 1)
    assert bench.auto_packing(2, 32, 768, False) == (2, 1)
    assert bench.auto_packing(16, 16, 256, False) == (3, 1)
    for k in range(1, 41):
        nf, vb = bench.auto_packing(k, 16, 512, False)
        assert vb == 1 or k % (nf * vb) == 0
