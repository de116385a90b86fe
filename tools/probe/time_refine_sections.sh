for S in 17 9 5; do export ES_REFINE_SECTIONS=$S; python -m pytest tests/test_shoot_gpu.py -q -m gpu -x -k "roots_vs_port" 2>&1 | tail -1; for ST in 1 2; do python bench.py --streams $ST --no-cpu-baseline --no-extra-mode --steps 30 > gpurun_out/sec_${S}_s$ST.json 2>/dev/null; done; python bench.py --share-of 8 --no-cpu-baseline --no-extra-mode --steps 60 > gpurun_out/sec_${S}_e8.json 2>/dev/null; python bench.py --workload config4 --no-cpu-baseline > gpurun_out/sec_${S}_c4.json 2>/dev/null; python - <<PY
import json
o=[]
for t in ("s1","s2","e8","c4"):
    j=json.loads(open("gpurun_out/sec_${S}_%s.json"%t).read().strip().splitlines()[-1]); o.append("%s %.3f ms (%.3e)"%(t,j["ms_per_step"],j["value"]))
print("sections $S:", "  ".join(o))
PY
done
