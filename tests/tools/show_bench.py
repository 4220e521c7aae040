"""Developer tool: one-line digest of bench.py JSON lines (step time, views/s, launch form, depth-limit counters, stage times).
    python tests/tools/show_bench.py gpurun_out/*.log"""
import json,sys
for f in sys.argv[1:]:
    j=json.loads(open(f).read().strip().splitlines()[-1])
    dl=j.get("depth_limit") or {}
    print(f, "ms/step %.4f"%j["ms_per_step"], "views/s %.1f"%j["value"], j.get("launch"), j.get("launch_trial"), dl.get("limited_views_in_timed_region"), dl.get("fallbacks_in_timed_region"), j["config"]["num_rendered_last_view"])
    print({k:round(v["ms_per_launch"],4) for k,v in j["stages"].items()})
