#!/bin/bash
# the four workloads' timed k_stream_reads dispatches under the SQ counters (the final build): tools/r4_sq_all.sh <tag>
tag=${1:-r4sq}
root=$(cd "$(dirname "$0")/.." && pwd)
$root/tools/sq_dispatch.sh $tag c2
$root/tools/sq_dispatch.sh $tag c3 --config C3
$root/tools/sq_dispatch.sh $tag c5 --config C5 --steps 17
$root/tools/sq_dispatch.sh $tag view --mode view
python3 $root/tools/resources.py > $root/gpurun_out/$tag/resources.txt
