#!/bin/bash
# Diagnostics (GPU box): where the host time of the config-3 training step goes.  bash tools/train_host_diag.sh > gpurun_out/<tag>/train_host_diag.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { echo "== $*"; python3 $R/tools/train_dp.py --steps 60 --warmup 10 "$@" 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.2f  clips/s %.1f  host %s" % (d["ms_per_step"], d["value"], str(d.get("host_ms_per_step")) + " " + str(d.get("allocator_per_step")) + " reserved MiB " + str(d.get("allocator_reserved_MiB"))))'; }
run --phase-times
run --phase-times --repeat-first-batch
run --phase-times --preload
export TTV_DIAG_WARM_PLANS=1
run --phase-times --preload
