timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_tile_tickets_gpu.py -q -x -k "split or tickets or drawn" 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --steps 200 --warmup 8 --no-train-object --no-cpu-baseline > gpurun_out/sw.json 2>gpurun_out/sw.err; python tools/bench_line.py gpurun_out/sw.json; done
python - <<EOF
import json
d=json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("group %.1f"%r["avg_us_per_launch_group"], ["%s %.1f"%(l["layers"],l["avg_us"]) for l in r["layers"]], "as_launched %.1f"%r["as_launched"]["avg_us_per_launch_group"])
EOF
