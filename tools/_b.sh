for c in 32 16; do RA_SPLIT_MIN_CIN=$c timeout 300 python bench.py --steps 200 --warmup 8 --no-train-object --no-cpu-baseline > gpurun_out/sw.json 2>gpurun_out/sw.err; echo "min cin $c: $(python tools/bench_line.py gpurun_out/sw.json)"
python - <<EOF
import json
d=json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("group %.1f"%r["avg_us_per_launch_group"], ["%s %.1f"%(l["layers"],l["avg_us"]) for l in r["layers"]], "as_launched %.1f"%r["as_launched"]["avg_us_per_launch_group"])
EOF
done
