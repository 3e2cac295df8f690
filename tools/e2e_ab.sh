#!/bin/bash
# end-to-end A/B of CLI variants on one generated input: tools/e2e_ab.sh <outdir> <bench args> -- prints wall seconds per variant
out=$1; shift; mkdir -p $out
run() { # name, env assignments...
  name=$1; shift
  env "$@" MM_TIMELINE=1 MM_LOADER_TIMING=1 MM_E2E_STDERR=$out/cli_$name.err timeout 900 python bench.py $BARGS > $out/e2e_$name.json 2> $out/e2e_$name.err
  python - $out/e2e_$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); g=d["gpu_cli"]
print("%-28s wall %.3f (first %.3f)  load %.3f  cpu_port %.3f  identical %s" % (sys.argv[2], g["wall_s"], g["wall_s_first_run"], g["stages_s"]["load"], d["cpu_port"]["wall_s"], d["parity_vs_cpu"]["byte_identical"]))
PY
}
BARGS="--e2e-gbases 12"
run plain A=1
run plain_skip MM_SKIP_TEARDOWN=1
run gpu MM_E2E_CLI_FLAGS=--gpu-inflate
run gpu_skip MM_E2E_CLI_FLAGS=--gpu-inflate MM_SKIP_TEARDOWN=1
BARGS="--config C3 --e2e-gbases 3"
run c3_plain A=1
run c3_gpu MM_E2E_CLI_FLAGS=--gpu-inflate
run c3_gpu_skip MM_E2E_CLI_FLAGS=--gpu-inflate MM_SKIP_TEARDOWN=1
