#!/bin/bash
# end of round 4 on the last build: the GPU suite, smoke(), the seven fuzz campaigns, the tied configuration end to end, 12 Gbases end to end
out=gpurun_out/${1:-r4v}; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $out/gpu_tests.txt; cat $out/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1 | tee $out/smoke.txt
tools/fuzz_campaigns.sh $out 300000 6 2>&1 | tail -24
timeout 600 tools/c3_replay_sweep.sh 2>&1 | tail -4; cp gpurun_out/c3replay/e2e_c3_3g.json $out/e2e_c3_3g.json
MM_E2E_STDERR=$out/e2e_c2_12g_cli_log.txt timeout 900 python bench.py --e2e-gbases 12 > $out/e2e_c2_12g.json 2> $out/e2e_c2_12g.err
python -c "
import json; d=json.loads(open('$out/e2e_c2_12g.json').read().strip().splitlines()[-1]); g=d['gpu_cli']; print('12G wall', g['wall_s'], g['stages_s'], 'cpu', d['cpu_port']['wall_s'], d['parity_vs_cpu']['byte_identical'])"
