#!/bin/bash
# end of round 3: the suites, the driver's bench line, and the 12-Gbases end-to-end figure with the CLI's defaults (device inflate by
# file size) and with --no-gpu-inflate
out=gpurun_out/r3zc; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json
MM_TIMELINE=1 MM_LOADER_TIMING=1 MM_E2E_STDERR=$out/cli_default.err timeout 900 python bench.py --e2e-gbases 12 > $out/e2e_c2_12g.json 2> $out/e2e_c2_12g.err
MM_TIMELINE=1 MM_LOADER_TIMING=1 MM_E2E_CLI_FLAGS=--no-gpu-inflate MM_E2E_STDERR=$out/cli_host_inflate.err timeout 900 python bench.py --e2e-gbases 12 > $out/e2e_c2_12g_host_inflate.json 2> $out/e2e_c2_12g_host_inflate.err
MM_LOADER_TIMING=1 timeout 900 python bench.py --config C3 --e2e-gbases 3 > $out/e2e_c3_3g.json 2> $out/e2e_c3_3g.err
python - $out/e2e_c2_12g.json $out/e2e_c2_12g_host_inflate.json $out/e2e_c3_3g.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); g=d["gpu_cli"]
    print("%s wall %.3f load %.3f cpu_port %.3f identical %s inflate: %s" % (f.split("/")[-1], g["wall_s"], g["stages_s"]["load"], d["cpu_port"]["wall_s"], d["parity_vs_cpu"]["byte_identical"], g.get("gpu_inflate")))
PY
