#!/bin/bash
# PMC passes + bench for the remaining bench modes so that their lines stop reporting pmc.stale
for m in ref c4 c5; do MODE=$m bash tools/collect_profiles.sh r04_final_$m > gpurun_out/collect_$m.log 2>&1; echo "collect $m rc=$?"; done
python - <<PY
import json
t=json.load(open("profiles/traffic.json")); print({k:(v.get("batch"), v.get("csrc_sha256","")[:8]) for k,v in t.items() if isinstance(v,dict)})
PY
cp profiles/traffic.json gpurun_out/traffic_all_modes.json
