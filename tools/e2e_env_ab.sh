#!/bin/bash
# same-box A/Bs of the 12-Gbase job under other environments: tools/e2e_env_ab.sh <tag> "<name:VAR=1 VAR2=2;name2:...>" [gbases] [config]
root=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; mkdir -p $root/gpurun_out/$tag
MM_TIMELINE=1 MM_E2E_ENV_VARIANTS="$2" python3 $root/bench.py --config ${4:-C2} --e2e-gbases ${3:-12} > $root/gpurun_out/$tag/e2e_env_ab.json 2> $root/gpurun_out/$tag/e2e_env_ab.err
python3 - <<PY
import json
d = json.load(open("$root/gpurun_out/$tag/e2e_env_ab.json"))
print("default wall %.3f" % d["gpu_cli"]["wall_s"], d["gpu_cli"]["stages_s"])
for name, v in d.get("env_variants", {}).items():
    for p in v["pairs"]:
        a, b = p["default"], p[name]
        print("%-10s default %.3f (load %.3f, outside %s)   %s %.3f (load %.3f, runtime %.3f, contexts %.3f, outside %s)" % (name, a["wall_s"], a["stages_s"].get("load", 0), a.get("outside"), name, b["wall_s"], b["stages_s"].get("load", 0), b["stages_s"].get("gpu_runtime", 0), b["stages_s"].get("contexts", 0), b.get("outside")))
    print("   same bytes as the CPU port:", v["byte_identical_to_cpu"])
PY
