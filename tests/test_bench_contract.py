"""The bench line's contract, checked on the line the final build printed on the GPU box (profiles/round6_final_bench.json,
copied there from the gpurun session): the keys the driver reads, the roofline and cpu_baseline objects, internal
consistency of the numbers.  bench.py itself needs a GPU; what it prints must not drift from what is documented."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line():
    return json.loads(open(os.path.join(ROOT, "profiles", "round6_final_bench.json")).read().strip().splitlines()[-1])


def test_driver_keys_and_types():
    d = line()
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (k, type(d[k]))
    assert d["vs_baseline"] is None and d["unit"] == "MB/s" and d["dtype"] == "u8" and d["scaling"] == "weak" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["bit_exact"] is True
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] or d["metric"].startswith(base["metric"][:30])


def test_numbers_are_consistent():
    d = line()
    n, ub = d["config"]["units_per_gpu"], d["config"]["unit_bytes"]
    assert abs(d["value"] - n * ub / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01          # value = bytes / time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["frac"] < 1.0
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02                                                   # the kernels fit into a step
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "MB/s" and c["sample"]


def test_host_inclusive_and_secondary_configs():
    d = line()
    h = d["host_inclusive"]
    assert h["bit_exact"] is True and "process" in h and h["MBps"] > 0 and h["to_host_MBps"] > 0
    assert h["MBps"] < d["value"] and h["to_host_MBps"] < h["MBps"]            # copies cost something; back to the host costs more
    assert "same_calls_from_this_process" in h
    sec = d["secondary"]
    assert len(sec) == 7 and all(s["bit_exact"] is True and s["roofline"]["frac"] < 1.0 for s in sec)
    # round 6: what real containers look like -- blocks that span frames, ONE long folder per codec with the
    # reference on one core beside it
    assert any("SPAN frames" in s["config"] and s["units_on_frame_parallel_path"] == 1.0 for s in sec)
    one = [s for s in sec if s["config"].startswith("ONE cabinet folder")]
    assert len(one) == 2 and all(s["cpu_baseline"]["cores"] == 1 and s["value"] > s["cpu_baseline"]["value"] for s in one)
    assert any("config 2" in s["config"] for s in sec) and any("config 3" in s["config"] for s in sec)
    assert any("config 4" in s["config"] for s in sec) and any("config 5" in s["config"] for s in sec)
    for s in sec:
        if "MSZIP" in s["config"] or "Quantum" in s["config"]:
            assert s["cpu_baseline"]["kind"] == "reference" and s["cpu_baseline"]["value"] > 0


def test_through_api_and_socket_estimate():
    """the product timed through its own object API (configs 2, 3, 4 as containers) in the line, with the
    time split and the bit-exactness of every extracted byte; the CPU ratio also against an estimated whole socket."""
    d = line()
    apis = [s["through_api"] for s in d["secondary"] if s.get("through_api")]
    assert len(apis) == 3
    for t in apis:
        assert t["bit_exact"] is True and t["errors"] == 0 and t["MBps"] > 0 and t["files"] > 0 and t["batch_calls"] >= 1
        assert abs(sum(t["split_ms"].values()) - t["seconds"] * 1e3) < 0.05 * t["seconds"] * 1e3 + 0.1     # the split adds up
        assert 0 < t["vs_host_inclusive_to_host"] < 1.0          # the object API cannot beat the batch ABI it sits on
    e = d["vs_cpu_baseline"]["single_socket_estimate"]
    assert e["physical_cores_per_socket"] >= 1 and "ESTIMATE" in e["what"]
    assert abs(e["MBps"] - d["cpu_baseline"]["one_core_MBps"] * e["physical_cores_per_socket"]) / e["MBps"] < 0.01
    assert abs(e["device_resident"] - d["value"] / e["MBps"]) < 0.02


def test_traffic_file_is_this_rounds():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert t["round"] == "round 6" and os.path.exists(os.path.join(ROOT, t["source"]))
    src = json.load(open(os.path.join(ROOT, t["source"])))
    assert src["fetch_kib_per_launch"] == t["fetch_kib_per_launch"] and src["write_kib_per_launch"] == t["write_kib_per_launch"]
    d = line()
    # (the line was printed minutes before the session's --pmc passes and replays the previous passes: within ten per cent)
    now = (2 * t["fetch_kib_per_launch"] + t["write_kib_per_launch"]) * 1024
    assert abs(d["roofline"]["traffic"] - now) / now < 0.10
    # SURVEY 8(d)'s metric as written, at the top level beside `value` (round 5)
    assert d["value_host_inclusive"] == d["host_inclusive"]["MBps"] and d["value_host_to_host"] == d["host_inclusive"]["to_host_MBps"]
    assert d["cpu_baseline"]["lscpu"].get("Core(s) per socket")
