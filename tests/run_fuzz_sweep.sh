#!/bin/bash
# The randomised sweeps of tests/test_gpu_fuzz.py at ten times (or $1 times) the suite's case count, on the GPU box:
#   bash tests/run_fuzz_sweep.sh [scale] [seed family] -> gpurun_out/fuzz_sweep.log, gpurun_out/fuzz_sweep_summary.json
# (lives under tests/ because it drives the oracle; nothing here is part of the product)
SCALE=${1:-10}
export NERF_AMD_FUZZ_SEED=${2:-0}
mkdir -p gpurun_out
NERF_AMD_FUZZ_SCALE=$SCALE NERF_AMD_QUIET=1 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider \
    --junitxml=gpurun_out/fuzz_sweep.xml > gpurun_out/fuzz_sweep.log 2>&1
rc=$?
python - <<'PY'
import glob, json, re, xml.etree.ElementTree as ET
root = ET.parse("gpurun_out/fuzz_sweep.xml").getroot()
suite = root if root.tag == "testsuite" else root.find("testsuite")
per = {}
for case in suite.iter("testcase"):
    name = re.sub(r"\[.*", "", case.get("name"))
    d = per.setdefault(name, {"cases": 0, "failed": 0, "skipped": 0, "seconds": 0.0, "failures": []})
    d["cases"] += 1
    d["seconds"] += float(case.get("time", 0))
    if case.find("failure") is not None or case.find("error") is not None:
        d["failed"] += 1
        d["failures"].append(case.get("name"))
    if case.find("skipped") is not None:
        d["skipped"] += 1
worst = {}
for f in glob.glob("gpurun_out/parity_fuzz_*.json"):
    v = json.load(open(f))
    kind = "fp32_split" if "fp32_split" in f else "fp32"
    w = worst.setdefault(kind, {"raw_max": 0.0, "rgb_max": 0.0, "z_well_max": 0.0, "z_max": 0.0, "cases": 0})
    w["cases"] += 1
    for k in ("raw_max", "rgb_max", "z_well_max", "z_max"):
        w[k] = max(w[k], float(v.get(k, 0.0)))
import os
out = {"family": int(os.environ.get("NERF_AMD_FUZZ_SEED", "0")), "tests": int(suite.get("tests")), "failures": int(suite.get("failures")) + int(suite.get("errors")),
       "skipped": int(suite.get("skipped")), "seconds": float(suite.get("time")), "per_sweep": per,
       "staged_worst_over_all_cases": worst}
for d in per.values():
    d["seconds"] = round(d["seconds"], 1)
json.dump(out, open("gpurun_out/fuzz_sweep_summary.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("tests", "failures", "skipped", "seconds")}))
PY
exit $rc
