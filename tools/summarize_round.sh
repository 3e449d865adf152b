#!/bin/bash
# After tools/round_end.sh: the committed summaries under profiles/ (usage: tools/summarize_round.sh r04)
R=${1:-r04}
python tools/summarize_profiles.py $R $R
for sc in helmet_syn lotus_syn monkey spheres; do python tools/summarize_profiles.py ${R}_$sc $R "" $sc; done
for f in scaling.txt update_probe.txt valu_issue.txt; do [ -s gpurun_out/$R/$f ] && cp gpurun_out/$R/$f profiles/${R}_$f; done
python - "$R" <<'PY'
import json, sys
R = sys.argv[1]
out = {}
for n in (2, 4):
    try: out[f"gloo_{n}_ranks_one_gpu"] = json.loads(open(f"gpurun_out/{R}/gloo_{n}.json").read().strip().splitlines()[-1])
    except Exception as e: out[f"gloo_{n}_ranks_one_gpu"] = {"error": str(e)}
try: out["one_process_2_handles_one_gpu"] = json.loads(open(f"gpurun_out/{R}/one_process.json").read().strip().splitlines()[-1])
except Exception as e: out["one_process_2_handles_one_gpu"] = {"error": str(e)}
json.dump(out, open(f"profiles/{R}_rehearsal_multi_rank_one_gpu.json", "w"), indent=1)
PY
ls profiles | grep "^$R" | tr '\n' ' '
