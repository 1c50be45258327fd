#!/bin/bash
# item 4: the multi-rank lane policy (4 lanes, the last one yields to a pending gather) forced on one GPU, 10 k steps
O=gpurun_out/r4dist; mkdir -p $O
A="--steps 10000 --warmup 100 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0"
python bench.py $A > $O/nodist.json 2> $O/nodist.err
VOG_BENCH_FORCE_DIST=1 VOG_FORCE_MULTI_RANK_LANES=1 python bench.py $A > $O/forced.json 2> $O/forced.err
VOG_BENCH_FORCE_DIST=1 python bench.py $A > $O/dist1.json 2> $O/dist1.err
python - <<'PY'
import json
r = {}
for n in ("nodist", "forced", "dist1"):
    d = json.loads(open(f"gpurun_out/r4dist/{n}.json").read().strip().splitlines()[-1])
    r[n] = d
    print(n, d["value"], d["ms_per_step"], d["parity"]["ok"], d["parity"]["non_finite_outputs_all_slots"], d["rccl_ranks"])
print("forced / nodist =", r["forced"]["value"] / r["nodist"]["value"])
json.dump({"steps": 10000, "no_dist": r["nodist"]["value"], "forced_multi_rank_lane_policy_one_rank_rccl_group": r["forced"]["value"],
           "one_rank_rccl_group_default_policy": r["dist1"]["value"], "ratio_forced_over_no_dist": r["forced"]["value"] / r["nodist"]["value"],
           "non_finite": r["forced"]["parity"]["non_finite_outputs_all_slots"], "parity_ok": r["forced"]["parity"]["ok"]},
          open("gpurun_out/r4dist/summary.json", "w"), indent=1)
PY
