# round 5, session h: the whole -m gpu suite as the driver runs it (-x), smoke, the default bench line
mkdir -p gpurun_out/r5h; O=gpurun_out/r5h
python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 > $O/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_full.log
( timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r5h/bench.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "value_host_inclusive", "value_host_to_host")}, d["roofline"]["frac"], d.get("vs_cpu_baseline", {}).get("device_resident"))
    for s in d.get("secondary", []):
        print(s.get("config", "")[:50], s.get("kernel_ms"), (s.get("through_api") or {}).get("MBps"))
    print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("lscpu"))
except Exception as e:
    print("bench parse failed", e)
P
