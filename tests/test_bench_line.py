"""bench.py's final stdout line: the contract's keys, built from canned measurements, must fit the driver's 8 000-character
tail of stdout + stderr (round 3's line was 24.5 KB and lost its headline there).  No GPU, no library."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _canned_det():
    roof = {"bound": "mfma", "kernel": "flash attention of the full-resolution blocks (down1/up5): N=65536 tokens, 4 heads, head_dim 16",
            "achieved": 716.7712345678, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.28670849382712,
            "traffic": 1356071701.3333333,
            "traffic_source": {"file": "profiles/r03_traffic.json", "commit": "4ff9f85", "kernel": "flash_bf16_nm<16, 4, 8, 0>",
                               "command": "python3 bench.py --steps 2 --warmup 1 --no-workloads", "src_hash": "137e9246817a2367", "stale": False},
            "binding_resource": "VALU issue (one v_exp_f32 per score, 64 FLOP per exp); priced against the bf16 MFMA peak -- DESIGN.md 4.1",
            "avg_launch_ms": 24.5432109876, "launches": 40, "forward_ms": 61.3456789012}
    return {"ms_per_step": 61.5912345678, "steps_timed": 20, "warmup": 5, "n_total": 16, "load_s": 0.612345, "dev_codec": True,
            "version": "ddpmir 0.4.0 src=137e9246817a2367 hipcc=7.2.26015-fc0010cf6a layout=3", "weight_blob_bytes": 702607104,
            "codec_threads": None, "roof": roof, "codec": None,
            "routing": {"general_kernel_workgroups_with_work": 0, "general_kernel_workgroups_launched": 81920, "share_general": 0.0,
                        "meaning": "query blocks whose score bound exceeds 80 (log2) take the running-max kernel"}}


def _canned_extras():
    out = []
    for w in bench.EXTRA_WORKLOADS:
        e = {"id": w["id"], "name": w["name"], "ms_per_step": 11.6123456789, "steps_per_sec": 86.123456, "images_per_sec": 0.68912345678,
             "roofline": {"frac": 0.2034567891, "kernel": "x" * 200}, "note": w.get("note")}
        if w.get("reference_it_per_s"):
            e["reference_it_per_s"] = w["reference_it_per_s"]
        out.append(e)
    out[3] = {"id": bench.EXTRA_WORKLOADS[3]["id"], "error": "RuntimeError('" + "y" * 400 + "')"}
    return out


def test_headline_line_fits_the_driver_tail_and_keeps_the_contract_keys():
    args = bench.parse_args(["--steps", "20", "--warmup", "5"])
    line = bench.headline_line(args, 1, "nccl", _canned_det(), 1, None)
    extras = _canned_extras()
    line["workloads_summary"] = bench.summarise_workloads(extras)
    line["workloads_file"] = {"path": "gpurun_out/bench_workloads.json", "child_rc": 0, "slices": len(extras)}
    line["general_kernel_only"] = {"ms_per_step": 68.8, "images_per_sec": 0.2325, "meaning": "same workload, every query block routed to the general (running-max) attention kernel"}
    line["cpu_baseline"] = {"value": 9.7313815436732e-05, "unit": "images/sec", "cores": 32, "kind": "port",
                            "sample": "numpy oracle, jpeg U-Net, B=1 256x256, 1 of 1000 timesteps, 10.28 s/step"}
    txt = bench.fit_line(line)
    assert "\n" not in txt and len(txt) < bench.LINE_LIMIT <= 6000, len(txt)
    d = json.loads(txt)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "library", "attention_routing", "general_kernel_only",
                "workloads_summary", "distributed"):
        assert key in d, key
    assert "dropped_for_length" not in d, d.get("dropped_for_length")
    assert len(d["workloads_summary"]) == len(bench.EXTRA_WORKLOADS)
    assert all(len(json.dumps(s)) < 160 for s in d["workloads_summary"])           # no names, no nested objects
    assert d["workloads_summary"][3]["error"] and d["workloads_summary"][2]["roofline_frac"] == 0.2035
    assert d["distributed"]["local_world_size"] == 1 and "codec_threads" in d["distributed"]
    assert abs(d["value"] - 16 / (1000 * d["ms_per_step"] / 1e3)) < 1e-12


def test_fit_line_drops_optional_keys_only():
    args = bench.parse_args([])
    line = bench.headline_line(args, 1, None, _canned_det(), 1, None)
    line["workloads_summary"] = [{"id": "w%d" % i, "blob": "z" * 300} for i in range(40)]      # 12 KB of optional payload
    line["cpu_baseline"] = {"value": 1e-4, "unit": "images/sec", "cores": 32, "kind": "port", "sample": "s"}
    d = json.loads(bench.fit_line(line))
    assert len(json.dumps(d)) < bench.LINE_LIMIT
    assert "workloads_summary" in d["dropped_for_length"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline"):
        assert key in d, key


def test_extra_workload_ids_are_short_and_unique():
    ids = [w["id"] for w in bench.EXTRA_WORKLOADS]
    assert len(set(ids)) == len(ids) and all(len(i) <= 12 for i in ids)
