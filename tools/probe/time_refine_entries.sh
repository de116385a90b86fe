#!/bin/bash
# GPU box: bracket refinement with node entries shared inside the wave (default) against per-lane entries
# (ES_REFINE_PRIVATE_ENTRIES=1): ms per step of the bench, one stream and pipelined, the 512-row tile, configs[4]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for V in shared private shared private; do
  if [ $V = private ]; then export ES_REFINE_PRIVATE_ENTRIES=1; else unset ES_REFINE_PRIVATE_ENTRIES; fi
  for ST in 1 2; do python $R/bench.py --streams $ST --no-cpu-baseline --no-extra-mode --steps 30 > $R/gpurun_out/ent_${V}_s$ST.json 2>/dev/null; done
  python $R/bench.py --share-of 8 --no-cpu-baseline --no-extra-mode --steps 60 > $R/gpurun_out/ent_${V}_e8.json 2>/dev/null
  python $R/bench.py --workload config4 --no-cpu-baseline > $R/gpurun_out/ent_${V}_c4.json 2>/dev/null
  python3 - <<PY
import json
o = []
for t in ("s1", "s2", "e8", "c4"):
    j = json.loads(open("$R/gpurun_out/ent_${V}_%s.json" % t).read().strip().splitlines()[-1]); o.append("%s %.3f ms" % (t, j["ms_per_step"]))
print("$V:", "  ".join(o))
PY
done
