#!/bin/bash
# a tied configuration (C3: two codes) end to end, canonical order against the replayed reference order, with other --gather values:
# tools/c3_replay_sweep.sh "<flags>" ...   ("" = the CLI's defaults)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/c3replay
[ $# -eq 0 ] && set -- ""
for g in "$@"; do
  MM_E2E_CLI_FLAGS="$g" MM_TIE_TIMING=1 MM_E2E_STDERR=gpurun_out/c3replay/c3_cli.txt timeout 900 python bench.py --config C3 --e2e-gbases 3 > gpurun_out/c3replay/e2e_c3_3g.json 2> gpurun_out/c3replay/e2e_c3_3g.err
  python -c "
import json; d=json.loads(open('gpurun_out/c3replay/e2e_c3_3g.json').read().strip().splitlines()[-1]); r=d['reference_order_replay']; print('flags [$g] canonical', round(d['gpu_cli']['wall_s'],3), 'cpu', round(d['cpu_port']['wall_s'],3), 'replay run wall', round(r['wall_s'],3), 'replay_s', r['replay_s'], 'wait', r['replay_waiting_for_gpu_s'], 'peak GB', r['peak_ram_gb'], 'same rows', r['same_rows_as_canonical'])"
  grep "mmh_tie_order_rows\|Real time\|GPU launches" gpurun_out/c3replay/c3_cli.txt.replay | cut -c1-200
done
